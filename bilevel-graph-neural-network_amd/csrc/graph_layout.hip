// COO (int64, reference layout) -> CSR (int32) and batch-vector -> segment offsets.
// Integer work, HBM/latency bound; no host synchronisation anywhere.
#include "common.hpp"

namespace blgnn {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

// counts[key+1]++ ; flag |= (input not sorted by (key, nbr))
__global__ void k_count(const int64_t *__restrict__ key, const int64_t *__restrict__ nbr,
                        int64_t nnz, int32_t *__restrict__ rowptr, int32_t *__restrict__ unsorted) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (; e < nnz; e += stride) {
        int64_t k = key[e];
        atomicAdd(&rowptr[k + 1], 1);
        if (e > 0) {
            int64_t pk = key[e - 1];
            if (pk > k || (pk == k && nbr[e - 1] > nbr[e])) bad = true;
        }
    }
    if (__any(bad) && (threadIdx.x & (WAVE - 1)) == 0) atomicOr(unsorted, 1);
}

__device__ __forceinline__ int block_inclusive_scan(int v, int *smem) {
    // wave scan then cross-wave
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    for (int o = 1; o < WAVE; o <<= 1) {
        int t = __shfl_up(v, o, WAVE);
        if (lane >= o) v += t;
    }
    if (lane == WAVE - 1) smem[wid] = v;
    __syncthreads();
    int add = 0;
    for (int w = 0; w < wid; ++w) add += smem[w];
    __syncthreads();
    return v + add;
}

// phase 1: per-tile inclusive scan in place, tile total -> sums[tile]
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_tiles(int32_t *__restrict__ a, int64_t len, int32_t *__restrict__ sums) {
    __shared__ int smem[SCAN_THREADS / WAVE];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS];
    int run = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        int64_t p = base + i;
        run += (p < len) ? a[p] : 0;
        v[i] = run;
    }
    int incl = block_inclusive_scan(run, smem);
    int off = incl - run;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        int64_t p = base + i;
        if (p < len) a[p] = v[i] + off;
    }
    if (threadIdx.x == SCAN_THREADS - 1) sums[blockIdx.x] = incl;
}

// phase 2: one block turns sums[] into an exclusive prefix (sequential chunks + carry)
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_sums(int32_t *__restrict__ sums, int ntiles) {
    __shared__ int smem[SCAN_THREADS / WAVE];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += SCAN_THREADS) {
        int p = base + threadIdx.x;
        int v = p < ntiles ? sums[p] : 0;
        int incl = block_inclusive_scan(v, smem);
        int carry = carry_s;
        if (p < ntiles) sums[p] = carry + incl - v;
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) carry_s = carry + incl;
        __syncthreads();
    }
}

// phase 3: add tile prefix
__global__ void __launch_bounds__(SCAN_THREADS)
k_scan_add(int32_t *__restrict__ a, int64_t len, const int32_t *__restrict__ sums) {
    int off = sums[blockIdx.x];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        int64_t p = base + i;
        if (p < len) a[p] += off;
    }
}

__global__ void k_copy_cursor(const int32_t *__restrict__ rowptr, int32_t n,
                              int32_t *__restrict__ cursor) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cursor[i] = rowptr[i];
}

// sorted input: col[e] = nbr[e]; otherwise claim a slot in the row
__global__ void k_fill(const int64_t *__restrict__ key, const int64_t *__restrict__ nbr,
                       int64_t nnz, int32_t *__restrict__ cursor, int32_t *__restrict__ col,
                       const int32_t *__restrict__ unsorted) {
    const bool sorted = (*unsorted == 0);
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; e < nnz; e += stride) {
        if (sorted) {
            col[e] = (int32_t)nbr[e];
        } else {
            int pos = atomicAdd(&cursor[key[e]], 1);
            col[pos] = (int32_t)nbr[e];
        }
    }
}

// unsorted input only: order each row ascending (deterministic summation order).
// One wave per row.  Rows of <= 64 entries: rank sort in registers.  Longer rows
// (rare; the interaction graph is laid out once per fold): in-place selection,
// one lane-parallel argmin per output position.
__global__ void k_sort_rows(const int32_t *__restrict__ rowptr, int32_t n,
                            int32_t *__restrict__ col, const int32_t *__restrict__ unsorted) {
    if (*unsorted == 0) return;
    const int lane = threadIdx.x & (WAVE - 1);
    int row = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
    const int row_step = gridDim.x * blockDim.x / WAVE;
    for (; row < n; row += row_step) {
        const int b = rowptr[row], len = rowptr[row + 1] - b;
        if (len <= 1) continue;
        if (len <= WAVE) {
            const int v = lane < len ? col[b + lane] : 0x7fffffff;
            int rank = 0;
            for (int j = 0; j < len; ++j) {
                int u = __shfl(v, j, WAVE);
                rank += (u < v) || (u == v && j < lane);
            }
            if (lane < len) col[b + rank] = v;
        } else {
            for (int p = 0; p < len - 1; ++p) {
                int best = 0x7fffffff, bestpos = 0x7fffffff;
                for (int j = p + lane; j < len; j += WAVE) {
                    int u = col[b + j];
                    if (u < best) { best = u; bestpos = j; }
                }
                for (int o = WAVE / 2; o > 0; o >>= 1) {
                    int ob = __shfl_xor(best, o, WAVE);
                    int op = __shfl_xor(bestpos, o, WAVE);
                    if (ob < best || (ob == best && op < bestpos)) { best = ob; bestpos = op; }
                }
                if (lane == 0 && bestpos != p) {
                    int t = col[b + p];
                    col[b + p] = best;
                    col[b + bestpos] = t;
                }
                __threadfence_block();
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

__global__ void k_segments(const int64_t *__restrict__ batch, int64_t n, int32_t G,
                           int32_t *__restrict__ seg) {
    // seg[g] = first i with batch[i] >= g  (batch sorted ascending)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    int64_t lo = (i == 0) ? 0 : batch[i - 1] + 1;      // graphs starting at or before i
    int64_t hi = (i == n) ? (int64_t)G : batch[i];
    for (int64_t g = lo; g <= hi && g <= G; ++g) seg[g] = (int32_t)i;
}

}  // namespace blgnn

using namespace blgnn;

extern "C" int blgnn_csr_from_coo(const int64_t *key, const int64_t *nbr, int64_t nnz,
                                  int32_t n, int32_t *rowptr, int32_t *col, int32_t *work,
                                  void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && nnz >= 0, "csr_from_coo: negative size");
    BLGNN_CHECK_ARG(rowptr && work, "csr_from_coo: null rowptr/work");
    BLGNN_CHECK_ARG(nnz == 0 || (key && nbr && col), "csr_from_coo: null edge arrays");
    BLGNN_CHECK_ARG(nnz < (int64_t)0x7fffffff, "csr_from_coo: nnz exceeds int32");
    hipStream_t s = (hipStream_t)stream;
    const int64_t len = (int64_t)n + 1;
    hipError_t me = hipMemsetAsync(rowptr, 0, sizeof(int32_t) * len, s);
    if (me == hipSuccess) me = hipMemsetAsync(work + n, 0, sizeof(int32_t), s);
    if (me != hipSuccess)
        return fail(BLGNN_ERR_LAUNCH, "csr_from_coo: hipMemsetAsync: %s", hipGetErrorString(me));
    if (nnz == 0 || n == 0) return launch_status("csr_from_coo(memset)");
    int32_t *flag = work + n;
    const int eb = (int)((nnz + 255) / 256 < 4096 ? (nnz + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_count, dim3(eb), dim3(256), 0, s, key, nbr, nnz, rowptr, flag);
    const int ntiles = (int)((len + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(SCAN_THREADS), 0, s, rowptr, len, work);
    if (ntiles > 1) {
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, s, work, ntiles);
        hipLaunchKernelGGL(k_scan_add, dim3(ntiles), dim3(SCAN_THREADS), 0, s, rowptr, len, work);
    }
    hipLaunchKernelGGL(k_copy_cursor, dim3((n + 255) / 256), dim3(256), 0, s, rowptr, n, work);
    hipLaunchKernelGGL(k_fill, dim3(eb), dim3(256), 0, s, key, nbr, nnz, work, col, flag);
    const int rb = (n + 3) / 4 < 2048 ? (n + 3) / 4 : 2048;
    hipLaunchKernelGGL(k_sort_rows, dim3(rb), dim3(256), 0, s, rowptr, n, col, flag);
    return launch_status("csr_from_coo");
}

extern "C" int blgnn_segments_from_batch(const int64_t *batch, int64_t n, int32_t G,
                                         int32_t *seg, void *stream) {
    BLGNN_CHECK_ARG(n >= 0 && G >= 0 && seg, "segments_from_batch: bad args");
    BLGNN_CHECK_ARG(n == 0 || batch, "segments_from_batch: null batch");
    hipStream_t s = (hipStream_t)stream;
    const int64_t threads = n + 1;
    hipLaunchKernelGGL(k_segments, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s,
                       batch, n, G, seg);
    return launch_status("segments_from_batch");
}
