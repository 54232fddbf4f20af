// gemm2s_probe -- the LATENCY-bound Linear launches of the MNIST step (1024 x 512 x 512: 3.4 us of matrix time inside a
// 12-14 us launch) on the LDS-DMA machinery of csrc/gemm2.h: one block per CU, k-grouped waves, the operands streamed
// global -> LDS by `buffer_load_dwordx4 ... lds` into a DEEP ring (no staging registers: the ring depth costs LDS only), a
// cooperative epilogue (k-group sum + bias + Swish, two outputs).  Timed as 20 launches inside a hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm2s_probe.hip -o tools/bin/gemm2s_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ void dma16(i32x4_t rs, int voff, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_byte) : "memory");
}
constexpr int BUF_OOB = (int)0x80000000u;
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ i32x4_t make_rsrc(const float *p, long floats, int records) {
    const unsigned long long a = (unsigned long long)p + (unsigned long long)floats * 4ull;
    i32x4_t r;
    r.x = uni((int)(unsigned)a); r.y = uni((int)((unsigned)(a >> 32) & 0xffffu)); r.z = uni(records); r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ float swishf_(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); }

// TMW x TNW tile waves (32 x 32 each) x KW k-groups; a barrier step is BK = 8 * KW * CH deep (CH chunks of 8 per group)
template <int TMW, int TNW, int KW, int CH, int S, bool Q_RK, int NIW>
__global__ __launch_bounds__(64 * TMW * TNW * KW)
void lat_gemm_kernel(const float *__restrict__ P, int ldp, const float *__restrict__ Q, int ldq, float *__restrict__ pre,
                     float *__restrict__ act, int ldd, const float *__restrict__ bias, int I, int J, int K) {
    constexpr int BM = 32 * TMW, BN = 32 * TNW, BK = 8 * KW * CH, NT = 64 * TMW * TNW * KW;
    constexpr int F = BK / 4;                                  // float4 slots per row of a k-contiguous image
    constexpr int P_FLOATS = BM * BK, Q_FLOATS = BN * BK, STAGE_FLOATS = P_FLOATS + Q_FLOATS;
    constexpr int NA = P_FLOATS / 256, NB = Q_FLOATS / 256;    // DMA instructions (1 KiB) per step
    static_assert(NA % NIW == 0 && NB % NIW == 0, "pieces must divide over the issuing waves");
    constexpr int NPW = (NA + NB) / NIW;
    static_assert(NPW * (S - 2) <= 63, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = uni(t >> 6);
    const int kg = wave / (TMW * TNW), wq = wave % (TMW * TNW), wi = wq / TNW, wj = wq % TNW;
    const int lrow = lane >> 5, lcol = lane & 31;
    // XCD-local sub-grids: XCD x (launch slots x, x + 8, ...) owns tiles_i / 8 row bands x all column tiles
    const int tiles_j = (J + BN - 1) / BN, tiles_i = (I + BM - 1) / BM;
    int b = blockIdx.x;
    {
        const int per = tiles_i * tiles_j / 8;
        if (tiles_i % 8 == 0) b = (b & 7) * per + (b >> 3);
    }
    const int ti = b / tiles_j, tj = b % tiles_j;
    const int i0 = ti * BM, j0 = tj * BN;
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void *)lds;
    auto swz = [](int r) { return F == 4 ? (r >> 2) & 3 : F == 8 ? (r >> 1) & 7 : r & 15; };

    // ---- DMA pieces of the issuing waves: piece q = wave + NIW * u  (u < NPW); q < NA: P piece, else Q piece q - NA
    int voff[NPW];
    if (wave < NIW) {
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int q = wave + NIW * u;
            if (NIW * u < NA) {                                   // compile-time per u (NA % NIW == 0)
                const int slot = q * 64 + lane, r = slot / F, f = (slot % F) ^ swz(r);
                voff[u] = (i0 + r < I) ? (r * ldp + f * 4) * 4 : BUF_OOB;
            } else if (Q_RK) {
                const int slot = (q - NA) * 64 + lane, r = slot / F, f = (slot % F) ^ swz(r);
                voff[u] = (j0 + r < J) ? (r * ldq + f * 4) * 4 : BUF_OOB;
            } else {
                const int slot = (q - NA) * 64 + lane, k = slot / (BN / 4), n = (slot % (BN / 4)) * 4;
                voff[u] = (j0 + n < J) ? (k * ldq + n) * 4 : BUF_OOB;
            }
        }
    }
    const float *Pb = P + (size_t)i0 * ldp, *Qb = Q_RK ? Q + (size_t)j0 * ldq : Q + j0;
    auto issue = [&](int kt, int stage) {
        if (wave >= NIW) return;
        const int k0 = kt * BK;
        const i32x4_t rp = make_rsrc(Pb, k0, 0x7fffffff);
        const i32x4_t rq = Q_RK ? make_rsrc(Qb, k0, 0x7fffffff) : make_rsrc(Qb, (long)k0 * ldq, 0x7fffffff);
        asm volatile("s_nop 4" ::: "memory");
#pragma unroll
        for (int u = 0; u < NPW; ++u) {
            const int q = wave + NIW * u;
            const bool isp = NIW * u < NA;
            dma16(isp ? rp : rq, voff[u], uni(lds0 + (stage * STAGE_FLOATS + (isp ? q * 256 : P_FLOATS + (q - NA) * 256)) * 4));
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int pbase = (wi * 32 + lcol) * BK;
    const int qbase = Q_RK ? (wj * 32 + lcol) * BK : 4 * lrow * BN + wj * 32 + lcol;
    const int fsw = swz(lcol);                                     // rows are 32-multiples + lcol: the swizzle is the lane's

    const int nk = (K + BK - 1) / BK;                              // (K % BK == 0 in this probe)
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < nk) issue(s, s);
    int st_c = 0, st_i = S - 1;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + S - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPW * (S - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + S - 1 < nk) { issue(kt + S - 1, st_i); st_i = st_i + 1 == S ? 0 : st_i + 1; }
        const float *Ps = lds + st_c * STAGE_FLOATS;
        const float *Qs = Ps + P_FLOATS;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int ch = kg * CH + c;                            // this group's chunk of 8 k's
            const float4 pa = *reinterpret_cast<const float4 *>(Ps + pbase + 4 * ((2 * ch + lrow) ^ fsw));
            float4 qb;
            if (Q_RK) qb = *reinterpret_cast<const float4 *>(Qs + qbase + 4 * ((2 * ch + lrow) ^ fsw));
            else qb = make_float4(Qs[qbase + (8 * ch + 0) * BN], Qs[qbase + (8 * ch + 1) * BN], Qs[qbase + (8 * ch + 2) * BN],
                                  Qs[qbase + (8 * ch + 3) * BN]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.x, qb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.y, qb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.z, qb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.w, qb.w, acc, 0, 0, 0);
        }
        st_c = st_c + 1 == S ? 0 : st_c + 1;
    }
    // ---- cooperative epilogue: park the k-groups' tiles, sum them in group order, bias + Swish, two outputs
    asm volatile("s_barrier" ::: "memory");
    constexpr int TP = BN + 1;
    float *tile = lds + kg * (BM * TP);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int il = wi * 32 + 4 * lrow + (r & 3) + 8 * (r >> 2);
        tile[il * TP + wj * 32 + lcol] = acc[r];
    }
    __syncthreads();
    for (int el = t; el < BM * BN; el += NT) {
        const int il = el / BN, jl = el % BN;
        float v = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < KW; ++g2) v += lds[g2 * (BM * TP) + il * TP + jl];
        const int i = i0 + il, j = j0 + jl;
        if (i < I && j < J) {
            v += bias[j];
            pre[(size_t)i * ldd + j] = v;
            act[(size_t)i * ldd + j] = swishf_(v);
        }
    }
}

template <bool Q_RK>
__global__ void ref_kernel(const float *P, int ldp, const float *Q, int ldq, const float *bias, double *D, int I, int J, int K, int stride) {
    const long o = (long)(blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (o >= (long)I * J) return;
    const int i = (int)(o / J), j = (int)(o % J);
    double s = bias[j];
    for (int k = 0; k < K; ++k) s += (double)P[(size_t)i * ldp + k] * (Q_RK ? Q[(size_t)j * ldq + k] : Q[(size_t)k * ldq + j]);
    D[o / stride] = s;
}

static float *dev_rand(size_t n, unsigned seed) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 32768.f - 1.f; }
    float *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <int TMW, int TNW, int KW, int CH, int S, bool Q_RK, int NIW>
static void run(const char *name, int I, int J, int K) {
    constexpr int BM = 32 * TMW, BN = 32 * TNW, BK = 8 * KW * CH, NT = 64 * TMW * TNW * KW;
    const int ldp = K, ldq = Q_RK ? K : J;
    float *P = dev_rand((size_t)I * K, 1), *Q = dev_rand((size_t)J * K, 2), *bias = dev_rand(J, 3), *pre, *act;
    CK(hipMalloc(&pre, (size_t)I * J * 4)); CK(hipMalloc(&act, (size_t)I * J * 4));
    auto kern = lat_gemm_kernel<TMW, TNW, KW, CH, S, Q_RK, NIW>;
    size_t lds_bytes = (size_t)S * (BM + BN) * BK * 4;
    const size_t red = (size_t)KW * BM * (BN + 1) * 4;
    if (red > lds_bytes) lds_bytes = red;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const int blocks = ((I + BM - 1) / BM) * ((J + BN - 1) / BN);
    hipStream_t st; CK(hipStreamCreate(&st));
    auto launch = [&]() { kern<<<blocks, NT, lds_bytes, st>>>(P, ldp, Q, ldq, pre, act, J, bias, I, J, K); };
    launch(); CK(hipStreamSynchronize(st));
    const int stride = 31;
    const long ns = ((long)I * J + stride - 1) / stride;
    double *R; CK(hipMalloc(&R, ns * 8));
    ref_kernel<Q_RK><<<(unsigned)((ns + 255) / 256), 256, 0, st>>>(P, ldp, Q, ldq, bias, R, I, J, K, stride);
    CK(hipStreamSynchronize(st));
    std::vector<double> hr(ns); std::vector<float> hd((size_t)I * J);
    CK(hipMemcpy(hr.data(), R, ns * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hd.data(), pre, (size_t)I * J * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (long s = 0; s < ns; ++s) { maxerr = fmax(maxerr, fabs(hr[s] - hd[s * stride])); maxref = fmax(maxref, fabs(hr[s])); }
    // 20 launches in a graph, replayed 5 times between two events
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int r = 0; r < 20; ++r) launch();
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(exec, st));
    float best = 1e30f;
    for (int round = 0; round < 3; ++round) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(exec, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / 100);
    }
    const double fl = 2.0 * I * J * K;
    printf("%-18s %3dx%-3d kw%d ch%d (BK %3d) S%d %s thr%4d lds%6zu blocks%4d | %7.2f us | %5.1f TFLOP/s | err %.2e / %.2e\n", name, BM, BN, KW,
           CH, BK, S, Q_RK ? "RK" : "MN", NT, lds_bytes, blocks, best * 1e3, fl / (best * 1e-3) / 1e12, maxerr, maxref);
    CK(hipFree(P)); CK(hipFree(Q)); CK(hipFree(pre)); CK(hipFree(act)); CK(hipFree(bias)); CK(hipFree(R));
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(st));
}

int main() {
    // MNIST B = 512, two passes stacked: 1024 x 512 x 512 forward (both operands k-contiguous) and data gradient (weights k-major)
    //   TMW TNW KW CH  S  RK   NIW
    run<1, 2, 4, 1, 4, true, 4>("mnist fwd", 1024, 512, 512);
    run<1, 2, 4, 1, 6, true, 4>("mnist fwd", 1024, 512, 512);
    run<1, 2, 4, 1, 8, true, 4>("mnist fwd", 1024, 512, 512);
    run<1, 2, 4, 2, 4, true, 4>("mnist fwd", 1024, 512, 512);
    run<1, 2, 4, 2, 6, true, 4>("mnist fwd", 1024, 512, 512);
    run<2, 1, 4, 1, 6, true, 4>("mnist fwd", 1024, 512, 512);
    run<1, 2, 8, 1, 4, true, 8>("mnist fwd", 1024, 512, 512);
    run<1, 2, 2, 2, 6, true, 4>("mnist fwd", 1024, 512, 512);
    run<2, 2, 2, 2, 4, true, 8>("mnist fwd", 1024, 512, 512);
    run<2, 2, 4, 1, 4, true, 8>("mnist fwd", 1024, 512, 512);
    run<1, 1, 4, 2, 6, true, 4>("mnist fwd", 1024, 512, 512);
    run<1, 1, 8, 1, 6, true, 4>("mnist fwd", 1024, 512, 512);
    run<1, 2, 4, 1, 6, false, 4>("mnist dgrad", 1024, 512, 512);
    run<1, 2, 4, 2, 4, false, 4>("mnist dgrad", 1024, 512, 512);
    run<1, 2, 4, 1, 6, true, 4>("mnist 784->512", 512, 512, 768);
    run<1, 1, 8, 1, 6, true, 4>("mnist M512", 512, 512, 512);
    run<1, 1, 4, 2, 6, true, 4>("mnist M512", 512, 512, 512);
    run<1, 2, 4, 1, 6, true, 4>("mnist 512->784", 1024, 768, 512);
    return 0;
}
