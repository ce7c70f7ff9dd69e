// bfk_sort.hip — stable LSD radix sort of (32-bit key, 32-bit row) records for the prefix-group path: the
// (max_dist + 2) * N records {prefix element : row * recs + slot} are brought into key order; a GROUP is a run of equal keys
// with its rows ascending (the sort is stable and the records are produced in row order).
//
// Hand-written for gfx950 (round 2 called rocPRIM's radix_sort_pairs here — three 8-bit passes for the 18 key bits of the
// benchmark vocabulary, ~200 us for 7M records).  The key bits that matter are few and known (bits of the largest token
// id + 2), so the digits are made as wide as they need to be for FEWER passes: ceil(bits / 11) passes of ceil(bits / passes)
// bits — two 9-bit passes for 18 bits.  Per pass:
//   k_rs_count    a block = 4096 consecutive records (8 waves of 512, 64 at a time: coalesced — 4 waves of 1024 left the scatter
//                 kernel, whose LDS allows 3 blocks per CU, at 12 waves per CU: 72 -> 58 us per pass with 24); digit histogram of
//                 the block in LDS -> hist[digit][block]
//   k_rs_rowscan  one wave per digit: exclusive prefix over the blocks, in place; the digit's total -> tot[digit]
//   k_rs_scatter  digit bases = exclusive scan of tot[] (in LDS, every block for itself); position of a record = digit base +
//                 records of that digit in earlier blocks + in earlier waves of the block + in earlier rounds of the wave +
//                 in lower lanes of the round (the lanes of a round that share a digit find each other with one ballot per
//                 digit bit) — stable by construction.  The block first orders its tile by digit in LDS and then writes it
//                 out run by run (each lane straight to its output position: 139 us per pass, 64 different lines per store;
//                 through the LDS tile: see DESIGN 6d)
// The input arrays stay intact (the walk reads the unsorted keys row by row); the passes ping-pong between the output
// arrays and a pair of temporaries so that the last pass writes the output.
#include <cstdint>
#include <cstring>

#include <hip/hip_runtime.h>

namespace bfk {

namespace {

constexpr int RS_THREADS = 512, RS_WAVES = 8, RS_KPL = 8;    // records per lane
constexpr int RS_WAVE_KEYS = 64 * RS_KPL;                    // 512 consecutive records per wave
constexpr int RS_TILE = RS_WAVES * RS_WAVE_KEYS;             // 4096 per block
constexpr int RS_MAX_BITS = 11;                              // widest digit: 2048 bins (2 x 32 KiB of LDS in the scatter kernel)

__device__ __forceinline__ int rs_wave_incl_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, true);  // row_bcast15 into rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, true);  // row_bcast31 into rows 2,3
    return x;
}

template <int RB>
__global__ __launch_bounds__(RS_THREADS) void k_rs_count(const uint32_t *__restrict__ keys, uint32_t n, int shift, uint32_t n_blk,
                                                         uint32_t *__restrict__ hist /*[1 << RB][n_blk]*/) {
    constexpr int R = 1 << RB;
    __shared__ uint32_t s_hist[R];
    for (int d = threadIdx.x; d < R; d += RS_THREADS) s_hist[d] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)RS_TILE + (threadIdx.x >> 6) * (uint32_t)RS_WAVE_KEYS + (threadIdx.x & 63);
#pragma unroll
    for (int j = 0; j < RS_KPL; j++) {
        const uint32_t i = base + j * 64u;
        if (i < n) atomicAdd(&s_hist[(keys[i] >> shift) & (R - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < R; d += RS_THREADS) hist[(size_t)d * n_blk + blockIdx.x] = s_hist[d];
}

// one wave per digit: exclusive prefix of the digit's counts over the blocks, in place; total -> tot[digit]
__global__ __launch_bounds__(64) void k_rs_rowscan(uint32_t *__restrict__ hist, uint32_t n_blk, uint32_t *__restrict__ tot) {
    uint32_t *row = hist + (size_t)blockIdx.x * n_blk;
    const int lane = threadIdx.x;
    uint32_t run = 0;
    // (four chunks of 64 requested together: the row is one dependent chain of loads otherwise — 27 round trips at 1M rows)
    for (uint32_t b0 = 0; b0 < n_blk; b0 += 256) {
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t b = b0 + u * 64 + lane;
            v[u] = b < n_blk ? (int)row[b] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t b = b0 + u * 64 + lane;
            const int inc = rs_wave_incl_scan(v[u]);
            if (b < n_blk) row[b] = run + (uint32_t)(inc - v[u]);
            run += (uint32_t)__builtin_amdgcn_readlane(inc, 63);
        }
    }
    if (lane == 0) tot[blockIdx.x] = run;
}

template <int RB>
__global__ __launch_bounds__(RS_THREADS) void k_rs_scatter(const uint32_t *__restrict__ keys, const int *__restrict__ vals, uint32_t n,
                                                           int shift, uint32_t n_blk, const uint32_t *__restrict__ hist,
                                                           const uint32_t *__restrict__ tot, uint32_t *__restrict__ keys_out,
                                                           int *__restrict__ vals_out, int comp_recs, int comp_pb, uint32_t gen_n, int gen_recs) {
    constexpr int R = 1 << RB;
    __shared__ uint32_t s_run[R];             // first OUTPUT position of the block's records of every digit
    __shared__ uint32_t s_loc[R];             // first position of the digit inside the block's (digit-ordered) tile
    __shared__ uint32_t s_w[RS_WAVES][R];     // per wave: its digit counts, then its running positions inside the tile
    __shared__ uint32_t s_part[RS_WAVES];
    __shared__ uint32_t s_key[RS_TILE];       // the tile in digit order: a digit's records leave as one contiguous run
    __shared__ int s_val[RS_TILE];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PER = R / RS_THREADS >= 1 ? R / RS_THREADS : 1;  // digits per thread: thread t owns [t * PER, (t + 1) * PER)
    for (int d = threadIdx.x; d < RS_WAVES * R; d += RS_THREADS) (&s_w[0][0])[d] = 0u;
    __syncthreads();
    // phase A: the wave's keys (kept in registers) and its digit counts
    const uint32_t tile0 = blockIdx.x * (uint32_t)RS_TILE;
    const uint32_t base = tile0 + wave * (uint32_t)RS_WAVE_KEYS + lane;
    uint32_t key[RS_KPL];
#pragma unroll
    for (int j = 0; j < RS_KPL; j++) {
        const uint32_t i = base + j * 64u;
        key[j] = i < n ? keys[i] : 0xFFFFFFFFu;
        if (i < n) atomicAdd(&s_w[wave][(key[j] >> shift) & (R - 1)], 1u);
    }
    __syncthreads();
    // phase B: two exclusive scans over the digits — of tot[] (records of smaller digits in the whole input) and of the block's
    // own counts (records of smaller digits in the tile) — and per digit the counts of the earlier waves in front of each wave
    {
        uint32_t gt[PER], bc[PER];
        uint32_t gsum = 0, bsum = 0;
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int d = threadIdx.x * PER + q;
            gt[q] = d < R ? tot[d] : 0u;
            bc[q] = 0u;
            if (d < R)
#pragma unroll
                for (int w = 0; w < RS_WAVES; w++) bc[q] += s_w[w][d];
            gsum += gt[q];
            bsum += bc[q];
        }
        const int ginc = rs_wave_incl_scan((int)gsum), binc = rs_wave_incl_scan((int)bsum);
        if (lane == 63) {
            s_part[wave] = (uint32_t)ginc;
            s_loc[wave] = (uint32_t)binc;  // (scratch until the barrier below)
        }
        __syncthreads();
        uint32_t grun = (uint32_t)ginc - gsum, brun = (uint32_t)binc - bsum;
        for (int w = 0; w < wave; w++) {
            grun += s_part[w];
            brun += s_loc[w];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const int d = threadIdx.x * PER + q;
            if (d < R) {
                s_run[d] = grun + hist[(size_t)d * n_blk + blockIdx.x];
                s_loc[d] = brun;
                uint32_t run = brun;
#pragma unroll
                for (int w = 0; w < RS_WAVES; w++) {
                    const uint32_t c = s_w[w][d];
                    s_w[w][d] = run;
                    run += c;
                }
            }
            grun += gt[q];
            brun += bc[q];
        }
    }
    __syncthreads();
    // phase C: every wave for itself, round by round (64 consecutive records): the lanes of a round with one digit find each
    // other by ballots; place in the tile = the wave's running place of the digit + lanes of the same digit below
    uint32_t *wrun = s_w[wave];
#pragma unroll
    for (int j = 0; j < RS_KPL; j++) {
        const uint32_t i = base + j * 64u;
        const bool valid = i < n;
        const uint32_t dg = (key[j] >> shift) & (R - 1);
        unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const bool bit = (dg >> b) & 1u;
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(bit);
            peers &= bit ? bal : ~bal;
        }
        const unsigned long long below = peers & ((1ull << lane) - 1ull);
        uint32_t pos = 0;
        if (valid) pos = wrun[dg] + (uint32_t)__popcll(below);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid && below == 0ull) wrun[dg] += (uint32_t)__popcll(peers);  // the lowest lane of the digit moves it on
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) {
            s_key[pos] = key[j];
            // (vals == NULL, first pass: the value of input position i is worked out — (i % gen_n) * gen_recs + i / gen_n for a
            // position-major input of gen_n rows, i itself otherwise — instead of being stored by the producer and loaded here)
            s_val[pos] = vals ? vals[i] : (gen_n ? (int)((i % gen_n) * (uint32_t)gen_recs + i / gen_n) : (int)i);
        }
    }
    __syncthreads();
    // phase D: the tile leaves in digit order — consecutive threads write consecutive output positions of a digit's run
    const uint32_t cnt = min((uint32_t)RS_TILE, n - tile0);
    for (uint32_t e = threadIdx.x; e < cnt; e += RS_THREADS) {
        const uint32_t k = s_key[e];
        const uint32_t dg = (k >> shift) & (R - 1);
        const uint32_t o = s_run[dg] + (e - s_loc[dg]);
        const int v = s_val[e];
        // (the last pass of the prefix-group sort: the key leaves as the composite {key : slot} the positional filter bisects
        // on — slot = value % recs; the SHORT key 0 and "no record" stay what they are)
        keys_out[o] = (comp_recs > 0 && k != 0u && k != 0xFFFFFFFFu) ? (k << comp_pb) | (uint32_t)(v % comp_recs) : k;
        vals_out[o] = v;
    }
}

template <int RB>
int rs_pass(const uint32_t *kin, const int *vin, uint32_t *kout, int *vout, uint32_t n, int shift, uint32_t n_blk, uint32_t *hist,
            uint32_t *tot, hipStream_t st, int comp_recs, int comp_pb, uint32_t gen_n, int gen_recs) {
    hipLaunchKernelGGL(k_rs_count<RB>, dim3(n_blk), dim3(RS_THREADS), 0, st, kin, n, shift, n_blk, hist);
    hipLaunchKernelGGL(k_rs_rowscan, dim3(1 << RB), dim3(64), 0, st, hist, n_blk, tot);
    hipLaunchKernelGGL(k_rs_scatter<RB>, dim3(n_blk), dim3(RS_THREADS), 0, st, kin, vin, n, shift, n_blk, hist, tot, kout, vout, comp_recs, comp_pb, gen_n, gen_recs);
    return (int)hipGetLastError();
}

}  // namespace

// temp == nullptr: only *temp_bytes is set.  Stable sort by key bits [0, bits); keys_in / rows_in stay intact.
// rows_in == NULL: the values are the input positions (gen_n = 0) or, for a position-major input of gen_n rows, (p % gen_n) * gen_recs +
// p / gen_n.  comp_recs > 0: keys_out holds (key << comp_pb) | (row value % comp_recs) for every key but 0 and 0xFFFFFFFF.
int sort_records(void *temp, size_t *temp_bytes, const uint32_t *keys_in, uint32_t *keys_out, const int *rows_in,
                 int *rows_out, size_t n, int bits, hipStream_t st, int comp_recs, int comp_pb, int gen_n, int gen_recs) {
    bits = bits < 1 ? 1 : (bits > 32 ? 32 : bits);
    const int passes = (bits + RS_MAX_BITS - 1) / RS_MAX_BITS;
    const int rb = (bits + passes - 1) / passes;  // 1 .. 11
    const uint32_t n_blk = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
    const size_t pair_bytes = ((n * 4 + 255) / 256) * 256;
    const size_t hist_bytes = (((size_t)(1u << RS_MAX_BITS) * (n_blk + 1) * 4 + 255) / 256) * 256;
    const size_t need = 2 * pair_bytes + hist_bytes + (size_t)(1u << RS_MAX_BITS) * 4;
    if (!temp) {
        *temp_bytes = need;
        return 0;
    }
    if (*temp_bytes < need) return (int)hipErrorInvalidValue;
    if (n == 0) return 0;
    if (n > 0xFFFF0000ull) return (int)hipErrorInvalidValue;
    char *tp = (char *)temp;
    uint32_t *tk = (uint32_t *)tp;
    int *tv = (int *)(tp + pair_bytes);
    uint32_t *hist = (uint32_t *)(tp + 2 * pair_bytes);
    uint32_t *tot = (uint32_t *)(tp + 2 * pair_bytes + hist_bytes);
    const uint32_t *kin = keys_in;
    const int *vin = rows_in;
    for (int p = 0; p < passes; p++) {
        // the last pass writes the output arrays: with an odd number of passes the first one does too
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        uint32_t *kout = to_out ? keys_out : tk;
        int *vout = to_out ? rows_out : tv;
        const bool last = p == passes - 1;
        int e = 0;
        switch (rb) {
            case 1: case 2: case 3: case 4: case 5: case 6: case 7: case 8:
                e = rs_pass<8>(kin, vin, kout, vout, (uint32_t)n, p * rb, n_blk, hist, tot, st, last ? comp_recs : 0, comp_pb, (uint32_t)gen_n, gen_recs); break;
            case 9: e = rs_pass<9>(kin, vin, kout, vout, (uint32_t)n, p * rb, n_blk, hist, tot, st, last ? comp_recs : 0, comp_pb, (uint32_t)gen_n, gen_recs); break;
            case 10: e = rs_pass<10>(kin, vin, kout, vout, (uint32_t)n, p * rb, n_blk, hist, tot, st, last ? comp_recs : 0, comp_pb, (uint32_t)gen_n, gen_recs); break;
            default: e = rs_pass<11>(kin, vin, kout, vout, (uint32_t)n, p * rb, n_blk, hist, tot, st, last ? comp_recs : 0, comp_pb, (uint32_t)gen_n, gen_recs); break;
        }
        if (e) return e;
        kin = kout;
        vin = vout;
    }
    return 0;
}

}  // namespace bfk
