// gemm2s_chain_probe -- what does a LAUNCH BOUNDARY between two dependent 1024 x 512 x 512 Linear + Swish layers cost, and what
// would an in-kernel hand-over cost instead?  The MNIST step is a chain of ~20 dependent launches of 8-12 us with 3.4 us of
// matrix time each.  Here L such layers run
//   (a) as L launches of the latency kernel (csrc/gemm2.h's gemm2s layout, as in tools/gemm2s_probe.hip), and
//   (b) as ONE launch: every block walks the L layers; layer l + 1's tile (row band ti) waits for a per-(layer, band) counter that
//       the tj-blocks of band ti bump after their epilogue stores (agent-scope release / acquire; the band's producers and its
//       consumers sit on the same XCD by the launch-order map, but nothing relies on that).  The weights of the next layer are
//       requested BEFORE the wait.  The last block to finish re-arms the counters (no memset node).
// Both inside a hipGraph (R repetitions of the L-layer chain), replayed between two events.  Outputs must be bit-identical.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm2s_chain_probe.hip -o tools/bin/gemm2s_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
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

constexpr int MAXL = 4;
struct ChainArgs {
    const float *in;                 // [I][K] input of layer 0
    const float *W[MAXL];            // [J][K] (k-contiguous rows)
    const float *bias[MAXL];
    float *pre[MAXL], *act[MAXL];    // [I][J]; act[l] is layer l + 1's input (J == K)
    int *cnt;                        // [MAXL][tiles_i] band counters, zero at launch, re-armed by the last block
    int *done;
    int first, L;                    // layers [first, first + L): L == 1 from a chain of launches = mode (a)
    int flags;                       // 1: hand-over by counters (mode b); 0: no waits (mode a, one layer per launch)
};

template <int TMW, int TNW, int KW, int CH, int S>
__global__ __launch_bounds__(64 * TMW * TNW * KW)
void chain_kernel(ChainArgs a, int I, int J, int K) {
    constexpr int NIW = 4;
    constexpr int BM = 32 * TMW, BN = 32 * TNW, BK = 8 * KW * CH, NT = 64 * TMW * TNW * KW;
    constexpr int F = BK / 4;
    constexpr int P_FLOATS = BM * BK, Q_FLOATS = BN * BK, STAGE_FLOATS = P_FLOATS + Q_FLOATS;
    constexpr int NA = P_FLOATS / 256, NB = Q_FLOATS / 256;
    static_assert(NA % NIW == 0 && NB % NIW == 0, "pieces must divide over the issuing waves");
    constexpr int NPA = NA / NIW, NPB = NB / NIW, NPW = NPA + NPB;
    static_assert(NPW * (S - 2) <= 63, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = uni(t >> 6);
    const int kg = wave / (TMW * TNW), wq = wave % (TMW * TNW), wi = wq / TNW, wj = wq % TNW;
    const int lrow = lane >> 5, lcol = lane & 31;
    const int tiles_j = (J + BN - 1) / BN, tiles_i = (I + BM - 1) / BM;
    int b = blockIdx.x;
    if (tiles_i % 8 == 0) b = (b & 7) * (tiles_i * tiles_j / 8) + (b >> 3);
    const int ti = b / tiles_j, tj = b % tiles_j;
    const int i0 = ti * BM, j0 = tj * BN;
    const unsigned lds0 = (unsigned)(unsigned long)(lds_void *)lds;
    auto swz = [](int r) { return F == 4 ? (r >> 2) & 3 : F == 8 ? (r >> 1) & 7 : r & 15; };
    int voffp[NPA], voffq[NPB];
    if (wave < NIW) {
#pragma unroll
        for (int u = 0; u < NPA; ++u) {
            const int q = wave + NIW * u, slot = q * 64 + lane, r = slot / F, f = (slot % F) ^ swz(r);
            voffp[u] = (i0 + r < I) ? (r * K + f * 4) * 4 : BUF_OOB;
        }
#pragma unroll
        for (int u = 0; u < NPB; ++u) {
            const int q = wave + NIW * u, slot = q * 64 + lane, r = slot / F, f = (slot % F) ^ swz(r);
            voffq[u] = (j0 + r < J) ? (r * K + f * 4) * 4 : BUF_OOB;
        }
    }
    const int pbase = (wi * 32 + lcol) * BK, qbase = (wj * 32 + lcol) * BK, fsw = swz(lcol);
    const int nk = (K + BK - 1) / BK;
    constexpr int TP = BN + 1;

    for (int l = a.first; l < a.first + a.L; ++l) {
        const float *Pb = (l == 0 ? a.in : a.act[l - 1]) + (size_t)i0 * K;
        const float *Qb = a.W[l] + (size_t)j0 * K;
        auto issue_q = [&](int kt, int stage) {
            if (wave >= NIW) return;
            const i32x4_t rq = make_rsrc(Qb, kt * BK, 0x7fffffff);
            asm volatile("s_nop 4" ::: "memory");
#pragma unroll
            for (int u = 0; u < NPB; ++u)
                dma16(rq, voffq[u], uni(lds0 + (stage * STAGE_FLOATS + P_FLOATS + (wave + NIW * u) * 256) * 4));
        };
        auto issue_p = [&](int kt, int stage) {
            if (wave >= NIW) return;
            const i32x4_t rp = make_rsrc(Pb, kt * BK, 0x7fffffff);
            asm volatile("s_nop 4" ::: "memory");
#pragma unroll
            for (int u = 0; u < NPA; ++u)
                dma16(rp, voffp[u], uni(lds0 + (stage * STAGE_FLOATS + (wave + NIW * u) * 256) * 4));
        };
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // the weights first: they do not depend on the previous layer
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < nk) issue_q(s, s);
        if (a.flags && l > a.first) {
            if (t == 0) {
                const int *c = a.cnt + (l - 1) * tiles_i + ti;
                int spins = 0;
                while (((a.flags & 4) ? __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : __hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < tiles_j) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1 << 22)) break;                 // never hang the box: a wrong result is caught by the check
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int s = 0; s < S - 1; ++s)
            if (s < nk) issue_p(s, s);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");         // the S - 1 prologue steps have landed
        int st_c = 0, st_i = S - 1;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt > 0) {
                if (kt + S - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" :: "n"(NPW * (S - 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            }
            if (kt + S - 1 < nk) { issue_p(kt + S - 1, st_i); issue_q(kt + S - 1, st_i); st_i = st_i + 1 == S ? 0 : st_i + 1; }
            const float *Ps = lds + st_c * STAGE_FLOATS;
            const float *Qs = Ps + P_FLOATS;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int ch = kg * CH + c;
                const float4 pa = *reinterpret_cast<const float4 *>(Ps + pbase + 4 * ((2 * ch + lrow) ^ fsw));
                const float4 qb = *reinterpret_cast<const float4 *>(Qs + qbase + 4 * ((2 * ch + lrow) ^ fsw));
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.x, qb.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.y, qb.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.z, qb.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa.w, qb.w, acc, 0, 0, 0);
            }
            st_c = st_c + 1 == S ? 0 : st_c + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        float *tile = lds + kg * (BM * TP);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int il = wi * 32 + 4 * lrow + (r & 3) + 8 * (r >> 2);
            tile[il * TP + wj * 32 + lcol] = acc[r];
        }
        __syncthreads();
        const float *bias = a.bias[l];
        float *pre = a.pre[l], *act = a.act[l];
        for (int el = t; el < BM * BN; el += NT) {
            const int il = el / BN, jl = el % BN;
            float v = 0.f;
#pragma unroll
            for (int g2 = 0; g2 < KW; ++g2) v += lds[g2 * (BM * TP) + il * TP + jl];
            const int i = i0 + il, j = j0 + jl;
            if (i < I && j < J) {
                v += bias[j];
                pre[(size_t)i * J + j] = v;
                if (a.flags & 4) __hip_atomic_store(act + (size_t)i * J + j, swishf_(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // write-through (sc1)
                else act[(size_t)i * J + j] = swishf_(v);
            }
        }
        if (a.flags & 1) {
            __threadfence();                                        // every thread's stores, agent scope
            __syncthreads();                                        // ... and the LDS tile is free for the next layer's ring
            if (t == 0 && l + 1 < a.first + a.L) __hip_atomic_fetch_add(a.cnt + l * tiles_i + ti, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (a.flags & 2) {
            __syncthreads();                                        // workgroup scope: the block's stores are out
            if (t == 0 && l + 1 < a.first + a.L) {
                __threadfence();                                    // ONE agent-scope release per block
                __hip_atomic_fetch_add(a.cnt + l * tiles_i + ti, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (a.flags & 4) {
            __syncthreads();                                        // every wave has waited for its (write-through) stores
            if (t == 0 && l + 1 < a.first + a.L) __hip_atomic_fetch_add(a.cnt + l * tiles_i + ti, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __syncthreads();
        }
    }
    if (a.flags && t == 0) {
        const int old = __hip_atomic_fetch_add(a.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (int)gridDim.x - 1) {                            // nobody polls any more: re-arm for the next launch
            for (int k = 0; k < MAXL * tiles_i; ++k) __hip_atomic_store(a.cnt + k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static float *dev_rand(size_t n, unsigned seed, float scale) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (((s >> 8) & 0xffff) / 32768.f - 1.f) * scale; }
    float *d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

template <int TMW, int TNW, int KW, int CH, int S>
static void run(int L, int I, int N, int R, int flags) {
    constexpr int BM = 32 * TMW, BN = 32 * TNW, BK = 8 * KW * CH, NT = 64 * TMW * TNW * KW;
    const int J = N, K = N;
    ChainArgs a; memset(&a, 0, sizeof(a));
    ChainArgs c; memset(&c, 0, sizeof(c));
    a.in = c.in = dev_rand((size_t)I * K, 1, 1.f);
    for (int l = 0; l < L; ++l) {
        a.W[l] = c.W[l] = dev_rand((size_t)J * K, 10 + l, 0.06f);
        a.bias[l] = c.bias[l] = dev_rand(J, 20 + l, 0.1f);
        CK(hipMalloc(&a.pre[l], (size_t)I * J * 4)); CK(hipMalloc(&a.act[l], (size_t)I * J * 4));
        CK(hipMalloc(&c.pre[l], (size_t)I * J * 4)); CK(hipMalloc(&c.act[l], (size_t)I * J * 4));
    }
    const int tiles_i = (I + BM - 1) / BM, tiles_j = (J + BN - 1) / BN;
    CK(hipMalloc(&c.cnt, (MAXL * tiles_i + 1) * 4)); CK(hipMemset(c.cnt, 0, (MAXL * tiles_i + 1) * 4));
    c.done = c.cnt + MAXL * tiles_i;
    auto kern = chain_kernel<TMW, TNW, KW, CH, S>;
    size_t lds_bytes = (size_t)S * (BM + BN) * BK * 4;
    const size_t red = (size_t)KW * BM * (BN + 1) * 4;
    if (red > lds_bytes) lds_bytes = red;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const int blocks = tiles_i * tiles_j;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, NT, lds_bytes));
    if (blocks > occ * 256) { printf("grid %d does not fit (occupancy %d per CU): no chain\n", blocks, occ); return; }
    hipStream_t st; CK(hipStreamCreate(&st));
    auto launch_a = [&]() {
        for (int l = 0; l < L; ++l) { ChainArgs x = a; x.first = l; x.L = 1; x.flags = 0; kern<<<blocks, NT, lds_bytes, st>>>(x, I, J, K); }
    };
    auto launch_c = [&]() { ChainArgs x = c; x.first = 0; x.L = L; x.flags = flags; kern<<<blocks, NT, lds_bytes, st>>>(x, I, J, K); };
    launch_a(); launch_c(); launch_c(); CK(hipStreamSynchronize(st));
    size_t bad = 0;
    std::vector<float> ha((size_t)I * J), hc((size_t)I * J);
    for (int l = 0; l < L; ++l) {
        CK(hipMemcpy(ha.data(), a.act[l], ha.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), c.act[l], hc.size() * 4, hipMemcpyDeviceToHost));
        bad += memcmp(ha.data(), hc.data(), ha.size() * 4) != 0;
    }
    float us[2];
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int r = 0; r < R; ++r) { if (mode) launch_c(); else launch_a(); }
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipGraphLaunch(exec, st));
        float best = 1e30f;
        for (int round = 0; round < 5; ++round) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(exec, st));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms / (5 * R));
        }
        us[mode] = best * 1e3f;
        CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    }
    // after the timed chain launches: still identical?
    CK(hipStreamSynchronize(st));
    for (int l = 0; l < L; ++l) {
        CK(hipMemcpy(ha.data(), a.act[l], ha.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hc.data(), c.act[l], hc.size() * 4, hipMemcpyDeviceToHost));
        bad += memcmp(ha.data(), hc.data(), ha.size() * 4) != 0;
    }
    printf("flags %d I %4d N %3d L %d  %3dx%-3d kw%d ch%d S%d blocks %3d (occ %d) | %d launches %7.2f us | one chain launch %7.2f us | per boundary %+6.2f us | %s\n",
           flags, I, N, L, BM, BN, KW, CH, S, blocks, occ, L, us[0], us[1], L > 1 ? (us[0] - us[1]) / (L - 1) : 0.f, bad ? "MISMATCH" : "identical");
    CK(hipStreamDestroy(st));
}

int main() {
    for (int flags = 1; flags <= 4; flags *= 2) {
        for (int L = 1; L <= 4; ++L) run<1, 2, 4, 1, 4>(L, 1024, 512, 8, flags);
        if (flags == 1) continue;
        for (int L = 2; L <= 4; L += 2) run<1, 1, 8, 1, 4>(L, 512, 512, 8, flags);
        for (int L = 2; L <= 4; L += 2) run<1, 2, 4, 1, 4>(L, 2048, 512, 8, flags);     // 512 blocks: two per CU
    }
    return 0;
}
