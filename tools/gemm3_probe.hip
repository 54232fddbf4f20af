// gemm3_probe -- WHERE the short-reduction launches of the round-6 GEMM core lose their time (K sweep, store knock-outs,
// de-phased block starts); the kernel is gemm2_probe's.  Was: stand-alone bench of the round-6 GEMM core ("v2"): 128 x 128 block tiles, 64 x 64 wave tiles
// (2 x 2 v_mfma_f32_32x32x2_f32 per k-pair), both operands streamed global -> LDS by `buffer_load ... lds`
// (no register hop, no ds_write), a 2- or 3-deep LDS ring, ONE raw s_barrier per k-tile with counted vmcnt waits.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm3_probe.hip -o tools/bin/gemm3_probe
// Run on the GPU box: tools/bin/gemm2_probe [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
__device__ void llvm_buf_load_lds(i32x4_t rsrc, lds_void *lds, int size, int voffset, int soffset, int imm, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");

// LDS-DMA hidden from hipcc: with the builtin the compiler waits vmcnt(0) before the next ds_read of the array (it knows the
// DMA writes LDS and cannot tell the ring stages apart), which drains the prefetch every k-tile.  M0 (the LDS destination) is
// written and restored inside the statement; the caller counts vmcnt by hand.
__device__ __forceinline__ void dma16(i32x4_t rs, int voff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_byte) : "memory");
}

constexpr int BUF_OOB = (int)0x80000000u;

struct SBase { unsigned lo, hi; };
__device__ __forceinline__ SBase sbase(const float *p) {
    const unsigned long long a = (unsigned long long)p;
    SBase b;
    b.lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    b.hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return b;
}
__device__ __forceinline__ i32x4_t make_rsrc(SBase b, long floats, int records) {
    const unsigned long long a = (((unsigned long long)b.hi << 32) | b.lo) + (unsigned long long)floats * 4ull;
    i32x4_t r;
    r.x = (int)(unsigned)a; r.y = (int)((unsigned)(a >> 32) & 0xffffu); r.z = records; r.w = 0x00020000;
    return r;
}

// Operand loader: a TILE x BK slab of S, streamed to LDS by LDS-DMA.
//   RK = true : S[r * ld + k] (reduction axis contiguous).  LDS image [TILE][BK], float4 slot f of row r stored at slot
//               f ^ swz(r) (XOR on the SOURCE address: the DMA destination is lane-linear), fragments by ds_read_b128.
//   RK = false: S[k * ld + r] (tile axis contiguous).  LDS image [BK][TILE], fragments by ds_read_b32.
template <bool RK, int BK, int TILE>
struct Op {
    static constexpr int F = BK / 4;                       // float4 per tile row (RK)
    static constexpr int NI = TILE * BK / 256;             // DMA instructions per tile (1 KiB each)
    static constexpr int V4 = TILE / 4;                    // float4 per k row (MN)
    static constexpr int NPW = NI / 4;                     // per wave
    static constexpr int FLOATS = TILE * BK;
    static_assert(NI % 4 == 0, "DMA instructions must divide over 4 waves");
    static __device__ __forceinline__ int swz(int r) { return BK == 16 ? (r >> 2) & 3 : BK == 32 ? (r >> 1) & 7 : r & 15; }
    int voff[NPW];
    SBase blk; int ld;
    __device__ void init(const float *S, int ld_, int tile0, int R, int lane, int wave) {
        ld = ld_;
        if (RK) {
            blk = sbase(S + (size_t)tile0 * ld);
#pragma unroll
            for (int u = 0; u < NPW; ++u) {
                const int slot = (wave * NPW + u) * 64 + lane;
                const int r = slot / F, fs = slot % F, f = fs ^ swz(r);
                voff[u] = (tile0 + r < R) ? (r * ld + f * 4) * 4 : BUF_OOB;
            }
        } else {
            blk = sbase(S + tile0);
#pragma unroll
            for (int u = 0; u < NPW; ++u) {
                const int slot = (wave * NPW + u) * 64 + lane;
                const int k = slot / V4, n = (slot % V4) * 4;
                voff[u] = (tile0 + n < R) ? (k * ld + n) * 4 : BUF_OOB;
            }
        }
    }
    // issue the DMA of the k-tile at k0 into the LDS image at byte address `dst`
    __device__ __forceinline__ void issue(unsigned dst, int k0, int kend, int wave) const {
        const i32x4_t rs = RK ? make_rsrc(blk, k0, 0x7fffffff)
                              : make_rsrc(blk, (long)k0 * ld, min((long)(kend - k0) * ld * 4, 0x7fffffffl));
#pragma unroll
        for (int u = 0; u < NPW; ++u) dma16(rs, voff[u], dst + (wave * NPW + u) * 1024);
    }
};

__device__ __forceinline__ int stag_mode_slots(int m) { return m < 0 ? -m : m; }

template <bool P_RK, bool Q_RK, int BK, int STAGES, int MINB, int WM, int WN>
__global__ __launch_bounds__(256, MINB)
void gemm2_kernel(const float *__restrict__ P, int ldp, const float *__restrict__ Q, int ldq, float *__restrict__ D, int ldd,
                  const float *__restrict__ bias, int I, int J, int K, int tiles_i, int tiles_j, int stag_mode, int stag_loops, int store_mode) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BM = 64 * WM, BN = 64 * WN;
    typedef Op<P_RK, BK, BM> OP;
    typedef Op<Q_RK, BK, BN> OQ;
    constexpr int STAGE_FLOATS = OP::FLOATS + OQ::FLOATS;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wi = wave >> 1, wj = wave & 1;
    const int lrow = lane >> 5, lcol = lane & 31;
    // XCD-aware tile order: launch slot b runs on XCD b % 8; XCD x owns a contiguous range of the tile list (i fastest)
    const int nt = tiles_i * tiles_j;
    int id = blockIdx.x;
    {
        const int q = nt >> 3, r = nt & 7, xcd = id & 7, slot = id >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int ti = id % tiles_i, tj = id / tiles_i;
    const int i0 = ti * BM, j0 = tj * BN;
    const int nk = (K + BK - 1) / BK;

    // de-phased starts: the blocks of the FIRST round (one per residency slot) wait slot * stag_loops * 8128 cycles
    if (stag_mode) {
        const int first = 256 * stag_mode_slots(stag_mode);
        if ((int)blockIdx.x < first) {
            const int slot = stag_mode > 0 ? (int)blockIdx.x / 256 : (int)blockIdx.x % stag_mode_slots(stag_mode);
            for (int w = 0; w < slot * stag_loops; ++w) __builtin_amdgcn_s_sleep(127);
        }
    }
    OP p; OQ q;
    p.init(P, ldp, i0, I, lane, wave);
    q.init(Q, ldq, j0, J, lane, wave);

    f32x16 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int b = 0; b < WN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // per-lane fragment addresses (float index inside an operand image)
    //   RK: row (w*64 + 32x + lcol), float4 slot (2c + lrow) ^ swz(row): swz depends on lcol only (64 and 32 are multiples of the period)
    //   MN: k row (2s + lrow), column w*64 + 32x + lcol
    //   (both: the j-th MFMA of chunk c sums k = 8c + j in lanes 0-31 and k = 8c + 4 + j in lanes 32-63)
    const int pbase = P_RK ? (wi * 32 * WM + lcol) * BK : 4 * lrow * BM + wi * 32 * WM + lcol;
    const int qbase = Q_RK ? (wj * 32 * WN + lcol) * BK : 4 * lrow * BN + wj * 32 * WN + lcol;
    const int pswz = OP::swz(lcol), qswz = OQ::swz(lcol);

    auto compute = [&](int stage) {
        const float *Ps = lds + stage * STAGE_FLOATS;
        const float *Qs = Ps + OP::FLOATS;
        float4 pa[WM], qb[WN], pa_n[WM], qb_n[WN];
        auto rd = [&](int c, float4 (&pa_)[WM], float4 (&qb_)[WN]) {
            if (P_RK) {
#pragma unroll
                for (int x = 0; x < WM; ++x) pa_[x] = *reinterpret_cast<const float4 *>(Ps + pbase + x * 32 * BK + 4 * ((2 * c + lrow) ^ pswz));
            }
            if (Q_RK) {
#pragma unroll
                for (int y = 0; y < WN; ++y) qb_[y] = *reinterpret_cast<const float4 *>(Qs + qbase + y * 32 * BK + 4 * ((2 * c + lrow) ^ qswz));
            }
        };
        rd(0, pa, qb);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            if (c + 1 < BK / 8) rd(c + 1, pa_n, qb_n);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float a[WM], b[WN];
#pragma unroll
                for (int x = 0; x < WM; ++x)
                    a[x] = P_RK ? (s == 0 ? pa[x].x : s == 1 ? pa[x].y : s == 2 ? pa[x].z : pa[x].w)
                                : Ps[pbase + (8 * c + s) * BM + x * 32];
#pragma unroll
                for (int y = 0; y < WN; ++y)
                    b[y] = Q_RK ? (s == 0 ? qb[y].x : s == 1 ? qb[y].y : s == 2 ? qb[y].z : qb[y].w)
                                : Qs[qbase + (8 * c + s) * BN + y * 32];
#pragma unroll
                for (int x = 0; x < WM; ++x)
#pragma unroll
                    for (int y = 0; y < WN; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[x], b[y], acc[x][y], 0, 0, 0);
            }
#pragma unroll
            for (int x = 0; x < WM; ++x) pa[x] = pa_n[x];
#pragma unroll
            for (int y = 0; y < WN; ++y) qb[y] = qb_n[y];
        }
    };
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void *)lds;
    auto issue = [&](int kt, int stage) {
        asm volatile("s_nop 4" ::: "memory");       // descriptor words may come fresh from readfirstlane
        p.issue(lds0 + stage * STAGE_FLOATS * 4, kt * BK, K, wave);
        q.issue(lds0 + (stage * STAGE_FLOATS + OP::FLOATS) * 4, kt * BK, K, wave);
    };
    constexpr int NPW = OP::NPW + OQ::NPW;      // DMA instructions per wave per k-tile

    if (STAGES == 3) {
        issue(0, 0);
        if (nk > 1) issue(1, 1);
        int kt = 0;
        // steady state: tiles kt (being waited for) and kt + 1 in flight; tile kt + 2 issued behind the barrier
        for (; kt + 2 < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPW) : "memory");
            issue(kt + 2, (kt + 2) % 3);
            compute(kt % 3);
        }
        if (kt + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPW) : "memory");
            compute(kt % 3);
            ++kt;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        compute(kt % 3);
    } else {
        issue(0, 0);
        int kt = 0;
        for (; kt + 1 < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            issue(kt + 1, (kt + 1) & 1);
            compute(kt & 1);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        compute(kt & 1);
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int y = 0; y < WN; ++y) {
        const int j = j0 + (wj * WN + y) * 32 + lcol;
        const float bj = (bias && j < J) ? bias[j] : 0.f;
#pragma unroll
        for (int x = 0; x < WM; ++x)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = i0 + (wi * WM + x) * 32 + 4 * lrow + (r & 3) + 8 * (r >> 2);
                if (store_mode == 1) { if (i < I && j < J && acc[x][y][r] == 12345.678f) D[(size_t)i * ldd + j] = acc[x][y][r] + bj; }
                else if (store_mode == 2) { if (i < I && j < J) __builtin_nontemporal_store(acc[x][y][r] + bj, &D[(size_t)i * ldd + j]); }
                else if (i < I && j < J) D[(size_t)i * ldd + j] = acc[x][y][r] + bj;
            }
    }
}

// reference: one thread per output, double accumulation
template <bool P_RK, bool Q_RK>
__global__ void ref_kernel(const float *P, int ldp, const float *Q, int ldq, double *D, int I, int J, int K, int stride) {
    const long o = (long)(blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (o >= (long)I * J) return;
    const int i = (int)(o / J), j = (int)(o % J);
    double s = 0;
    for (int k = 0; k < K; ++k) {
        const float a = P_RK ? P[(size_t)i * ldp + k] : P[(size_t)k * ldp + i];
        const float b = Q_RK ? Q[(size_t)j * ldq + k] : Q[(size_t)k * ldq + j];
        s += (double)a * b;
    }
    D[o / stride] = s;
}

static float *dev_rand(size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
    float *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <bool P_RK, bool Q_RK, int BK, int STAGES, int MINB, int WM = 2, int WN = 2>
static void run(const char *name, int I, int J, int K, int reps, int stag_mode = 0, int stag_loops = 0, int store_mode = 0) {
    const int ldp = P_RK ? K : I, ldq = Q_RK ? K : J;
    float *P = dev_rand((size_t)I * K, 1), *Q = dev_rand((size_t)J * K, 2), *D;
    CK(hipMalloc(&D, (size_t)I * J * 4));
    CK(hipMemset(D, 0xff, (size_t)I * J * 4));
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int ti = (I + BM - 1) / BM, tj = (J + BN - 1) / BN;
    auto kern = gemm2_kernel<P_RK, Q_RK, BK, STAGES, MINB, WM, WN>;
    const size_t lds_bytes = (size_t)STAGES * (BM + BN) * BK * 4;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds_bytes));
    auto launch = [&]() { kern<<<ti * tj, 256, lds_bytes, 0>>>(P, ldp, Q, ldq, D, J, nullptr, I, J, K, ti, tj, stag_mode, stag_loops, store_mode); };
    launch(); CK(hipDeviceSynchronize());
    // check a strided sample against the double reference
    const int stride = 97;
    const long ns = ((long)I * J + stride - 1) / stride;
    double *R; CK(hipMalloc(&R, ns * 8));
    ref_kernel<P_RK, Q_RK><<<(unsigned)((ns + 255) / 256), 256>>>(P, ldp, Q, ldq, R, I, J, K, stride);
    std::vector<double> hr(ns); std::vector<float> hd((size_t)I * J);
    CK(hipMemcpy(hr.data(), R, ns * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd.data(), D, (size_t)I * J * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    if (store_mode != 1) for (long s = 0; s < ns; ++s) { maxerr = fmax(maxerr, fabs(hr[s] - hd[s * stride])); maxref = fmax(maxref, fabs(hr[s])); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) launch();
    float best = 1e30f, sum = 0;
    for (int round = 0; round < 3; ++round) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / reps); sum += ms / reps;
    }
    const double fl = 2.0 * I * J * K;
    printf("%-14s stag %2d x%2d st%d %3dx%-3d I%5d J%5d K%5d BK%2d S%d minb%d occ%d blocks%5d | %8.2f us (best) %8.2f avg | %6.1f TFLOP/s = %.3f | err %.2e / %.2e\n",
           name, stag_mode, stag_loops, store_mode, BM, BN, I, J, K, BK, STAGES, MINB, occ, ti * tj, best * 1e3, sum / 3 * 1e3, fl / (best * 1e-3) / 1e12, fl / (best * 1e-3) / 157.3e12,
           maxerr, maxref);
    CK(hipFree(P)); CK(hipFree(Q)); CK(hipFree(D)); CK(hipFree(R));
}

int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    // (1) K sweep at the FashionMNIST decoder's shape: T(K) = a + b K
    for (int K : {128, 256, 512, 1024, 2048, 4096}) {
        run<true, true, 16, 3, 4, 1, 1>("fm fwd 64x64", 2048, 6272, K, reps);
        run<true, true, 16, 3, 2, 2, 1>("fm fwd 128x64", 2048, 6272, K, reps);
        run<true, true, 16, 3, 2>("fm fwd 128x128", 2048, 6272, K, reps);
    }
    // (2) the epilogue stores: off / non-temporal
    for (int sm : {1, 2}) {
        run<true, true, 16, 3, 4, 1, 1>("fm fwd 64x64", 2048, 6272, 512, reps, 0, 0, sm);
        run<true, true, 16, 3, 2, 2, 1>("fm fwd 128x64", 2048, 6272, 512, reps, 0, 0, sm);
        run<true, true, 16, 3, 2>("fm fwd 128x128", 2048, 6272, 512, reps, 0, 0, sm);
    }
    // (3) de-phased starts: slot = block / 256 (mode > 0) or block % slots (mode < 0), slots = blocks per CU
    for (int loops : {1, 2, 3, 4}) {
        run<true, true, 16, 3, 4, 1, 1>("fm fwd 64x64", 2048, 6272, 512, reps, 6, loops);
        run<true, true, 16, 3, 4, 1, 1>("fm fwd 64x64", 2048, 6272, 512, reps, -6, loops);
        run<true, true, 16, 3, 2, 2, 1>("fm fwd 128x64", 2048, 6272, 512, reps, 4, 2 * loops);
        run<true, true, 16, 3, 2, 2, 1>("fm fwd 128x64", 2048, 6272, 512, reps, -4, 2 * loops);
        run<true, true, 16, 3, 2>("fm fwd 128x128", 2048, 6272, 512, reps, 3, 4 * loops);
        run<true, true, 16, 3, 2>("fm fwd 128x128", 2048, 6272, 512, reps, -3, 4 * loops);
    }
    return 0;
}
