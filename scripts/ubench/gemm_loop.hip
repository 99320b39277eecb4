// GEMM k-loop microbenchmark (round 6): the LDS-DMA ring + fragment-read + MFMA loop of k_gemm8 (dn_gemm_kernels.h) as a stand-alone,
// CORRECT bf16 linear (C[M][N] = A[M][K] W[N][K]^T, K % 64 == 0, M % BM == 0, N % BN == 0), parameterised by
//   WM x WN waves per workgroup, wave tile (16 MT) x (16 NT), NS ring stages of 64 k -- to measure what wave tile / wave count the loop
// wants (LDS bytes per MFMA: (MT + NT) 1-KB fragment reads per MT NT MFMAs; DMA bytes per MFMA: (BM + BN) / (BM BN)).
// ABL bit 0: no DMA in the steady state, bit 1: no fragment reads in the steady state (results wrong: timing ablations only).
// build: hipcc --offload-arch=gfx950 -O3 -o gemm_loop gemm_loop.hip     run: ./gemm_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <type_traits>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ void glds16_s(const void *sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_s_nt(const void *sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_s_sc1(const void *sbase, unsigned voff, unsigned lds_dst)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned short f2bf(float f)
{
    unsigned u = __builtin_bit_cast(unsigned, f);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int WM, int WN, int MT, int NT, int NS, int ABL, int ORD = 0, int WNT = 0>
__global__ __launch_bounds__(64 * WM * WN, 1) void kg(const unsigned short *A, const unsigned short *W, unsigned short *C, int M, int N, int K)
{
    constexpr int NW = WM * WN, BM = 16 * MT * WM, BN = 16 * NT * WN;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int AG = BM / 8, WGp = BN / 8;
    constexpr int NPC = (AG + WGp + NW - 1) / NW, NMM = MT * NT;      // ragged split: the surplus pieces re-fetch group 0 into a 1-KB dummy slot behind the ring
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int nbn = N / BN;
    // XCD-contiguous tile order: workgroup b runs on XCD b % 8; logical tile = (b % 8) * (grid / 8) + b / 8 (grid % 8 == 0)
    const int lt = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int mblk = lt / nbn, nblk = lt % nbn;
    const int64_t m_base = (int64_t)mblk * BM, n_base = (int64_t)nblk * BN;
    const int nk = K / 64;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int lr = lane >> 3, ls = lane & 7;
    unsigned g_off[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        int g = wid + NW * i;                                         // 8-row group: A groups first, then W groups (wave-uniform)
        if (g >= AG + WGp) g = 0;
        const int row = (g < AG ? g : g - AG) * 8 + lr;
        const int ldA = ORD ? K / 9 : K;
        g_off[i] = (unsigned)((((g < AG ? m_base : n_base) + row) * (g < AG ? ldA : K) + (ls ^ ((row >> 1) & 7)) * 8) * 2);
    }
    const unsigned char *Ab = (const unsigned char *)A, *Wb = (const unsigned char *)W;
    auto issue_piece = [&](int kt, int stage, int p) __attribute__((always_inline)) {
        const int g = wid + NW * p;
        if (g >= AG + WGp) { glds16_s(Ab + (size_t)kt * 128, g_off[p], lds0 + NS * STAGE); return; }
        if constexpr (ORD == 0) { glds16_s((g < AG ? Ab : Wb) + (size_t)kt * 128, g_off[p], lds0 + stage * STAGE + (unsigned)(g * 1024)); return; }
        const int Cin = K / 9, ncs = Cin / 64;
        const int tap = ORD == 1 ? kt / ncs : kt % 9, cs = ORD == 1 ? kt % ncs : kt / 9;
        const int dy = tap / 3, dx = tap - 3 * dy;
        if (g < AG) glds16_s(Ab + ((size_t)(dy * 64 + dx) * Cin + cs * 64) * 2, g_off[p], lds0 + stage * STAGE + (unsigned)(g * 1024));
        else if (WNT == 1) glds16_s_nt(Wb + ((size_t)tap * Cin + cs * 64) * 2, g_off[p], lds0 + stage * STAGE + (unsigned)(g * 1024));
        else if (WNT == 2) glds16_s_sc1(Wb + ((size_t)tap * Cin + cs * 64) * 2, g_off[p], lds0 + stage * STAGE + (unsigned)(g * 1024));
        else glds16_s(Wb + ((size_t)tap * Cin + cs * 64) * 2, g_off[p], lds0 + stage * STAGE + (unsigned)(g * 1024));
    };
    auto issue = [&](int kt, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NPC; ++p) issue_piece(kt, stage, p);
    };
    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fc = lane >> 4;
    const int swz = (fr >> 1) & 7;
    const int fx0 = ((fc ^ swz) << 4), fx1 = (((fc + 4) ^ swz) << 4);
    const int aw0 = BM * 128 + (wn * (16 * NT) + fr) * 128, aa0 = (wm * (16 * MT) + fr) * 128;
    struct Frag { uint4 w[NT], a[MT]; };
    auto load_frag = [&](Frag &f, int stage, int ks) __attribute__((always_inline)) {
        const unsigned char *sb = smem + stage * STAGE;
        const int fx = ks ? fx1 : fx0;
#pragma unroll
        for (int t = 0; t < NT; ++t) f.w[t] = *reinterpret_cast<const uint4 *>(sb + aw0 + fx + t * 2048);
#pragma unroll
        for (int t = 0; t < MT; ++t) f.a[t] = *reinterpret_cast<const uint4 *>(sb + aa0 + fx + t * 2048);
    };
    auto block = [&](const Frag &f, Frag &fn, int st_next, int ks_next, bool load_next, int dma_kt, int dma_stage, auto dma_tag) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma_tag)::value;
        static_for<0, NMM>([&](auto m_) __attribute__((always_inline)) {
            constexpr int m = decltype(m_)::value, nt = m / MT, mt = m % MT;
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.w[nt]), __builtin_bit_cast(bf16x8, f.a[mt]), acc[nt][mt], 0, 0, 0);
            if constexpr (m == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (load_next) load_frag(fn, st_next, ks_next);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (DMA && m >= 2) {
                static_for<0, NPC>([&](auto p_) __attribute__((always_inline)) {
                    constexpr int pp = decltype(p_)::value;
                    if constexpr (m == 2 + (pp * (NMM - 3)) / NPC) {
                        __builtin_amdgcn_sched_barrier(0);
                        issue_piece(dma_kt, dma_stage, pp);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            }
        });
    };
    constexpr int GRPW = NPC;
    static_assert((NS - 2) * GRPW < 64, "vmcnt range");
    auto wait_tiles = [&](auto n_) __attribute__((always_inline)) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(n_)::value * GRPW) : "memory"); };
    Frag f0, f1;
#pragma unroll
    for (int s = 0; s < NS; ++s) if (s < nk) issue(s, s);
    wait_tiles(std::integral_constant<int, NS - 1>{});       // (nk >= NS assumed)
    __builtin_amdgcn_s_barrier();
    load_frag(f0, 0, 0);
    int st = 0, kt = 0;
    for (; kt + NS < nk; ++kt) {
        const int st1 = st + 1 == NS ? 0 : st + 1;
        block(f0, f1, st, 1, !(ABL & 2), 0, 0, std::false_type{});
        wait_tiles(std::integral_constant<int, NS - 2>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (ABL & 1) block(f1, f0, st1, 0, !(ABL & 2), 0, 0, std::false_type{});
        else block(f1, f0, st1, 0, !(ABL & 2), kt + NS, st, std::true_type{});
        st = st1;
    }
    for (; kt < nk; ++kt) {
        const int st1 = st + 1 == NS ? 0 : st + 1;
        block(f0, f1, st, 1, true, 0, 0, std::false_type{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        block(f1, f0, st1, 0, kt + 1 < nk, 0, 0, std::false_type{});
        st = st1;
    }
    const int64_t n_lane = n_base + wn * (16 * NT) + fc * 4;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = m_base + wm * (16 * MT) + mt * 16 + fr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int64_t n = n_lane + nt * 16;
            uint2 pk;
            pk.x = (unsigned)f2bf(acc[nt][mt][0]) | ((unsigned)f2bf(acc[nt][mt][1]) << 16);
            pk.y = (unsigned)f2bf(acc[nt][mt][2]) | ((unsigned)f2bf(acc[nt][mt][3]) << 16);
            *reinterpret_cast<uint2 *>((unsigned char *)C + (m * N + n) * 2) = pk;
        }
    }
}

static unsigned short f2bf_host(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int WM, int WN, int MT, int NT, int NS, int ABL, int ORD = 0, int WNT = 0>
void run(const char *name, int tiles_m, int tiles_n, int K, bool check)
{
    constexpr int BM = 16 * MT * WM, BN = 16 * NT * WN, STAGE = (BM + BN) * 128;
    const int M = BM * tiles_m, N = BN * tiles_n;
    static unsigned short *A = nullptr, *W = nullptr, *C = nullptr;
    static std::vector<unsigned short> hA, hW;
    const size_t cap = (size_t)288 << 20, capw = (size_t)8 << 20, capc = (size_t)32 << 20;
    if (!A) {
        hipMalloc(&A, cap * 2); hipMalloc(&W, capw * 2); hipMalloc(&C, capc * 2);
        hA.resize(cap); hW.resize(capw);
        for (size_t i = 0; i < cap; ++i) hA[i] = f2bf_host((float)((int)((i * 2654435761u >> 13) % 7) - 3));
        for (size_t i = 0; i < capw; ++i) hW[i] = f2bf_host((float)((int)((i * 40503u >> 7) % 5) - 2));
        hipMemcpy(A, hA.data(), cap * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), capw * 2, hipMemcpyHostToDevice);
    }
    if ((size_t)M * K > cap || (size_t)N * K > capw || (size_t)M * N > capc) { printf("%s: too big\n", name); return; }
    auto kern = kg<WM, WN, MT, NT, NS, ABL, ORD, WNT>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, NS * STAGE + 1024);
    const int grid = tiles_m * tiles_n;
    auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WM * WN), NS * STAGE + 1024, 0, A, W, C, M, N, K); };
    launch(); launch();
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: launch failed: %s\n", name, hipGetErrorString(e)); return; }
    double maxerr = -1;
    if (check && ABL == 0 && ORD == 0) {
        std::vector<unsigned short> hC((size_t)M * N);
        hipMemcpy(hC.data(), C, hC.size() * 2, hipMemcpyDeviceToHost);
        maxerr = 0;
        for (int s = 0; s < 4000; ++s) {
            const int m = (int)((s * 7919u + (s % 3) * (M - 1)) % M), n = (int)((s * 104729u + (s % 5) * (N - 1)) % N);
            double r = 0;
            for (int k = 0; k < K; ++k) r += (double)bf2f(hA[(size_t)m * K + k]) * bf2f(hW[(size_t)n * K + k]);
            const double d = fabs(r - bf2f(hC[(size_t)m * N + n]));
            const double tol = fabs(r) / 128.0 + 1e-3;
            if (d / tol > maxerr) maxerr = d / tol;
        }
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, tf = 2.0 * M * N * (double)K / (us * 1e-6) / 1e12;
    printf("%-34s tile %3dx%3d waves %dx%d wave-tile %3dx%3d NS=%d abl=%d ord=%d wnt=%d | M=%6d N=%5d K=%5d grid=%4d | %8.1f us %7.1f TF/s (%.3f of 2.5 PF) lds=%3d KB reads/mfma=%.2f %s\n",
           name, BM, BN, WM, WN, 16 * MT, 16 * NT, NS, ABL, ORD, WNT, M, N, K, grid, us, tf, tf / 2500.0, NS * STAGE / 1024, (double)(MT + NT) / (MT * NT),
           maxerr < 0 ? "" : (maxerr <= 1.0 ? "OK" : "MISMATCH"));
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int K = argc > 1 ? atoi(argv[1]) : 5760;
    if (argc > 2) {      // conv access pattern (64-wide image, Cin = K / 9): tap-outer vs tap-inner k order, W through L1 or not
        run<4, 2, 4, 4, 3, 0, 0, 0>("linear 256x128", 128, 2, K, true);
        run<4, 2, 4, 4, 3, 0, 1, 0>("conv tap-outer 256x128", 128, 2, K, false);
        run<4, 2, 4, 4, 3, 0, 2, 0>("conv tap-inner 256x128", 128, 2, K, false);
        run<4, 2, 4, 4, 3, 0, 2, 1>("conv tap-inner W nt", 128, 2, K, false);
        run<4, 2, 4, 4, 3, 0, 2, 2>("conv tap-inner W sc1", 128, 2, K, false);
        run<4, 2, 4, 4, 3, 0, 1, 1>("conv tap-outer W nt", 128, 2, K, false);
        run<4, 2, 4, 4, 3, 1, 0, 0>("no DMA", 128, 2, K, false);
        run<4, 2, 3, 4, 3, 0, 1, 0>("conv tap-outer 192x128", 128, 2, K, false);
        run<4, 2, 3, 4, 3, 0, 2, 0>("conv tap-inner 192x128", 128, 2, K, false);
        run<4, 2, 3, 4, 3, 0, 2, 1>("conv tap-inner W nt 192x128", 128, 2, K, false);
        run<4, 2, 3, 4, 3, 0, 2, 2>("conv tap-inner W sc1 192x128", 128, 2, K, false);
        run<4, 2, 3, 4, 3, 1, 0, 0>("no DMA 192x128", 128, 2, K, false);
        return 0;
    }
    // 8 waves (2 per SIMD, <= 256 registers): the product kernel's shape and wider wave tiles
    run<4, 2, 4, 4, 3, 0>("8w 64x64 (k_gemm8 MT=4 NTW=4)", 128, 2, K, true);
    run<4, 2, 4, 4, 3, 1>("  no DMA", 128, 2, K, false);
    run<4, 2, 4, 4, 3, 2>("  no frag reads", 128, 2, K, false);
    run<4, 2, 4, 4, 3, 3>("  neither", 128, 2, K, false);
    run<4, 2, 4, 8, 2, 0>("8w 64x128", 128, 2, K, true);
    run<4, 2, 4, 8, 2, 1>("  no DMA", 128, 2, K, false);
    run<4, 2, 4, 8, 2, 2>("  no frag reads", 128, 2, K, false);
    run<4, 2, 4, 8, 2, 3>("  neither", 128, 2, K, false);
    run<2, 4, 8, 4, 2, 0>("8w 128x64", 128, 2, K, true);
    run<4, 2, 6, 5, 2, 0>("8w 96x80 (384x160)", 128, 2, K, true);
    run<4, 2, 3, 4, 3, 0>("8w 48x64 (192x128)", 128, 2, K, true);
    run<4, 2, 4, 6, 2, 0>("8w 64x96 (256x192)", 128, 2, K, true);
    // 4 waves (1 per SIMD, <= 512 registers)
    run<2, 2, 6, 10, 2, 0>("4w 96x160", 256, 1, K, true);
    run<2, 2, 6, 10, 2, 1>("  no DMA", 256, 1, K, false);
    run<2, 2, 6, 10, 2, 2>("  no frag reads", 256, 1, K, false);
    run<2, 2, 6, 10, 2, 3>("  neither", 256, 1, K, false);
    run<2, 2, 8, 8, 2, 0>("4w 128x128", 128, 2, K, true);
    run<2, 2, 6, 5, 3, 0>("4w 96x80 (192x160)", 128, 2, K, true);
    run<1, 4, 6, 5, 3, 0>("4w 96x80 (96x320)", 256, 1, K, true);
    run<2, 2, 8, 4, 3, 0>("4w 128x64", 128, 2, K, true);
    run<2, 2, 4, 8, 3, 0>("4w 64x128", 128, 2, K, true);
    run<2, 2, 4, 4, 3, 0>("4w 64x64 (128x128)", 128, 2, K, true);
    return 0;
}
