"""Throughput of text steps with 1 .. 4 contexts (distributed.TextPipeline) and the host's share: time to ENQUEUE the steps
against time until they are done.  usage (GPU box): [GPU_MAX_HW_QUEUES=n] python tools/two_ctx.py [rows [text copies [idle context 0|1 [depths, e.g. 3,2,4]]]]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from breakfast_amd import _lib  # noqa: E402
from breakfast_amd.distributed import TextPipeline  # noqa: E402
from breakfast_amd.synth import generate_profiles  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
N_COPIES = int(sys.argv[2]) if len(sys.argv) > 2 else 12
IDLE_CTX = int(sys.argv[3]) if len(sys.argv) > 3 else 0
DEPTHS = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [3, 2]
rows = list(dict.fromkeys(generate_profiles(n)))
buf, off = _lib.pack_rows(rows)
T, n_u = len(buf), len(rows)
need = _lib.text_device_bytes(T)
h = torch.frombuffer(bytearray(buf), dtype=torch.uint8)
texts = []
for _ in range(N_COPIES):
    t = torch.empty(need, dtype=torch.uint8, device="cuda")
    t[:T].copy_(h)
    texts.append(t)
d_off = torch.from_numpy(off).cuda()
torch.cuda.synchronize()
ref = None
idle = None
if IDLE_CTX:
    from breakfast_amd.distributed import GpuEngine, ShardedClusterer
    idle = ShardedClusterer(GpuEngine(0), 0, 1)
    for _ in range(50):
        idle.step_text(texts[0].data_ptr(), T, d_off.data_ptr(), n_u, " ", 1)
    idle.e.sync()
for n_ctx in DEPTHS:
    pipe = TextPipeline(0, n_ctx)
    labs = [torch.empty(n_u, dtype=torch.int32, device="cuda") for _ in range(n_ctx)]
    k = [0]

    def step():
        i = k[0]
        k[0] += 1
        pipe.step_text(texts[i % N_COPIES].data_ptr(), T, d_off.data_ptr(), n_u, " ", 1, labs[i % n_ctx], inputs_ready=True, want_event=False)

    for _ in range(300):
        step()
    pipe.sync()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        t0 = time.perf_counter()
        K = 1000
        for _ in range(K):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append(((t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3))
    st = pipe.sync()
    assert st["n_retry_slices"] == 0
    lab = labs[0].cpu().numpy()
    if ref is None:
        ref = lab
    print(f"{n_ctx} context(s): host enqueue / done, ms per step: " + ", ".join(f"{a:.4f} / {b:.4f}" for a, b in res) +
          f"; labels equal: {np.array_equal(lab, ref)}", flush=True)
    pipe.close()
