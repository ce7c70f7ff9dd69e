"""Summarise gpurun_out/pf_waves.txt (BFK_PF_DEBUG=4 dump of k_prefilter's per-wave stamps, 100 MHz clock)."""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pf_waves.txt", dtype=np.uint64)
a = a[a[:, 1] > 0]
t0 = a[:, 1].min()
st, rng, mn, en = [(a[:, i].astype(np.int64) - int(t0)) / 100.0 for i in (1, 2, 3, 4)]
chunks, hits, nrows = a[:, 5].astype(int), a[:, 6].astype(int), a[:, 7].astype(int)
print(f"waves {len(a)}  kernel span {en.max():.1f} us  last start {st.max():.1f} us")
print(f"per wave: range look-up {np.mean(rng - st):.2f} us, scan {np.mean(mn - rng):.2f} us, flush {np.mean(en - mn):.2f} us, total mean {np.mean(en - st):.2f} max {np.max(en - st):.2f}")
print("chunks/wave mean %.1f max %d; hits/wave mean %.1f max %d" % (chunks.mean(), chunks.max(), hits.mean(), hits.max()))
for q in (50, 90, 99, 100):
    print(f"  p{q}: start {np.percentile(st, q):6.1f}  end {np.percentile(en, q):6.1f}  dur {np.percentile(en - st, q):6.1f}  scan {np.percentile(mn - rng, q):6.1f}  flush {np.percentile(en - mn, q):6.1f}")
o = np.argsort(-(en - st))[:8]
for i in o:
    print(f"  longest: tile-wave {int(a[i, 0])} start {st[i]:.1f} dur {en[i] - st[i]:.1f} scan {mn[i] - rng[i]:.1f} flush {en[i] - mn[i]:.1f} chunks {chunks[i]} hits {hits[i]} rows {nrows[i]}")

# per-SIMD load (HW_ID / XCC_ID recorded in column 8)
w = a[:, 8]
hw = ((w >> np.uint64(32)) & np.uint64(0xFFFF)).astype(int)
xcc = ((w >> np.uint64(48)) & np.uint64(0xF)).astype(int)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
sid = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
u, inv = np.unique(sid, return_inverse=True)
nw = np.bincount(inv)
ch = np.bincount(inv, weights=chunks)
busy_end = np.zeros(len(u)); np.maximum.at(busy_end, inv, en)
print(f"SIMDs used {len(u)}; waves/SIMD mean {nw.mean():.1f} min {nw.min()} max {nw.max()}; chunks/SIMD mean {ch.mean():.1f} min {ch.min():.0f} max {ch.max():.0f}")
print(f"SIMD end time: mean {busy_end.mean():.1f} p90 {np.percentile(busy_end, 90):.1f} max {busy_end.max():.1f}; corr(chunks, end) {np.corrcoef(ch, busy_end)[0, 1]:.2f}")
cuid = sid // 4
uc, invc = np.unique(cuid, return_inverse=True)
chc = np.bincount(invc, weights=chunks)
print(f"CUs used {len(uc)}; chunks/CU mean {chc.mean():.1f} min {chc.min():.0f} max {chc.max():.0f}; XCC counts {np.bincount(xcc)}")
