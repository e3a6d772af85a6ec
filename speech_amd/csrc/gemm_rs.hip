// Role-split persistent bf16 GEMMs for the SRU projections of net.py:61-72 and the classifier of
// net.py:73-76 at training / bulk-inference sizes.
//
//   NT:  C[M][N] = A[M][K] * B[N][K]^T     (both operands K-contiguous: forward U = LN(x) W on the packed
//        W^T, data gradient gX = gU W^T, classifier logits and their data gradient)
//
// One workgroup per CU, 12 waves with FIXED ROLES:
//   * waves 0-7 ("compute", two per SIMD) own the MFMA work: fragments LDS -> registers by ds_read_b128,
//     v_mfma_f32_16x16x32_bf16, and the epilogue stores.  They never wait on a vector-memory counter:
//     their only vector-memory instructions are the C stores, which drain in the background while the
//     next tile's K steps run.
//   * waves 8-11 ("loaders", one per SIMD) own every global -> LDS transfer: LDS-DMA
//     (global_load_lds_dwordx4) of whole 128-byte operand lines into a ring of K = 64 stages, a counted
//     s_waitcnt vmcnt, and nothing else.  On gfx950 loads and stores retire in issue order per wave, so
//     a wave that both stores C and waits for its next operand stage stalls on the stores'
//     acknowledgements; with the roles split that wait only ever covers loads, and no MFMA wave spends
//     issue slots on DMA instructions.
// One s_barrier per K step hands a landed stage to the compute waves and a consumed stage back to the
// loaders.  The workgroup walks its tiles as ONE flat sequence of K steps (the loaders run ahead across
// tile boundaries).  LDS rows are whole 128-byte lines with 16-byte chunk c of row r stored at chunk
// c ^ ((r >> 1) & 7): conflict-free for the 16-lane groups of ds_read_b128; LDS-DMA writes lane-linear,
// so the permutation is applied to the per-lane SOURCE address.  The MFMA takes the B rows as its row
// operand and the A rows as its column operand: a lane ends up with 4 consecutive n of one m; two
// v_permlane16_swap per pair of accumulators widen that to 16-byte stores (64 contiguous bytes per row
// and instruction) without an LDS round trip.  Tile order is XCD-aware (see tile ownership below).
#include "gemm.h"
#include <utility>

namespace sp {

typedef __bf16 rs_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rs_f16x8 __attribute__((ext_vector_type(8)));
typedef float rs_f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t rs_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* rs_gptr_t;
typedef __attribute__((address_space(3))) void* rs_lptr_t;

template <int N> struct IC { static constexpr int v = N; };
template <int... Is, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(IC<Is>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// One instruction each, in program order (asm volatile statements keep their order): LDS fragment read (the
// compiler does not know that it completes later -- every use is behind an RS_WAIT), counted wait, MFMA.
#define RS_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define RS_WAIT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n))
#define RS_MFMA(acc, p, q)                                                                                              \
    do {                                                                                                                \
        if constexpr (IN == SP_F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(p), "v"(q)); \
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(p), "v"(q));                      \
    } while (0)

struct RsArgs {
    const __hip_bfloat16* A; const __hip_bfloat16* B; void* C;
    int M, N, K, lda, ldb;
    int64_t ldc;
    int ntn, ntm, gcols, cgrp;
};

// XCD-aware tile ownership.  The eight XCDs (workgroup b runs on XCD b % 8) form a gcols x (8 / gcols)
// grid over the tile matrix: an XCD owns ntn / gcols whole tile COLUMNS of a contiguous range of tile
// rows, so its share of B stays resident in its 4 MB L2 and every A panel it streams is used by all of
// its columns back to back.  Its workgroups take the tiles of that sub-matrix round-robin: groups of cgrp
// columns, within a group columns fastest, then rows, then the next group.
struct RsTiles {
    int cols_per, col0, row_lo, run1, first, nslots, count, cgrp, per_grp;
    __device__ __forceinline__ RsTiles(int ntn, int ntm, int gcols, int cgrp_) {
        const int xcd = blockIdx.x & 7;
        nslots = gridDim.x >> 3;
        cols_per = ntn / gcols;
        const int grows = 8 / gcols;
        col0 = (xcd % gcols) * cols_per;
        row_lo = (int)((int64_t)(xcd / gcols) * ntm / grows);
        const int row_hi = (int)((int64_t)(xcd / gcols + 1) * ntm / grows);
        run1 = (row_hi - row_lo) * cols_per;
        cgrp = cgrp_;
        per_grp = (row_hi - row_lo) * cgrp;
        first = blockIdx.x >> 3;
        count = first < run1 ? (run1 - first + nslots - 1) / nslots : 0;
    }
    __device__ __forceinline__ int tile_m(int i) const { return row_lo + (i % per_grp) / cgrp; }
    __device__ __forceinline__ int tile_n(int i) const { return col0 + (i / per_grp) * cgrp + i % cgrp; }
};

// BM x BN tile, WMW x WNW compute waves (= 8), OUT = SP_BF16 or SP_F32.  The A rows and the B rows of a K = 64
// step live in two separate rings: B -- the weight panels, L2-resident, short latency -- two stages deep, A --
// activations or their gradients, streamed from beyond the L2 -- NSTA = 2 or 3 stages deep.  Every loader wave
// issues its share of B for the next step first and then its share of A for the step after (NSTA = 3); its
// counted wait before the barrier leaves exactly that A share in flight (a wave's vector-memory operations retire
// in issue order, so everything older has landed).  All four loaders share both operands: the far operand's
// requests leave in a quarter of the time one wave would need to issue them.
// IN = SP_BF16 or SP_F16: element type of A and B (and of C unless OUT == SP_F32)
template <int BM, int BN, int WMW, int NSTA, int OUT, int IN = SP_BF16>
__global__ __launch_bounds__(768, 3) void gemm_nt_rs_kernel(RsArgs g) {
    constexpr int WNW = 8 / WMW, NSTB = 2;
    constexpr int TM = BM / WMW, TN = BN / WNW;          // wave tile
    constexpr int MB = TM / 16, NB = TN / 16;
    constexpr int ROWB = 128, STA = BM * ROWB, STB = BN * ROWB, BOFF = NSTA * STA;
    constexpr int PA = BM / 32, PB = BN / 32;            // 8-row DMA groups per stage and loader wave
    static_assert(BM % 32 == 0 && BN % 32 == 0 && NB % 2 == 0 && TM % 16 == 0 && TN % 16 == 0, "tile shape");
    static_assert(NSTA == 2 || NSTA == 3, "A ring depth");
    static_assert(NSTA * STA + NSTB * STB <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTA * STA + NSTB * STB];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const RsTiles tl(g.ntn, g.ntm, g.gcols, g.cgrp);
    if (tl.count == 0) return;
    const int nk = g.K / 64;
    const int total = tl.count * nk;

    if (wave >= 8) {
        // ------------------------------------------------------------------ loader
        const int lw = wave - 8;
        const int drow = lane >> 3;
        const int lchunk = (lane & 7) ^ (((lw & 1) << 2) | (drow >> 1));      // row = 8 (lw + 4 i) + drow
        int a_off[PA], b_off[PB];
        // two cursors over the flat (tile, k) sequence: A runs one step further ahead than B when NSTA = 3
        int atile = tl.first, ak = 0, btile = tl.first, bk = 0;
        auto set_a = [&](int tile) {
            const int m0 = tl.tile_m(tile) * BM;
#pragma unroll
            for (int i = 0; i < PA; ++i)     // rows past M are never stored: clamp their source
                a_off[i] = min(m0 + 8 * (lw + 4 * i) + drow, g.M - 1) * g.lda + lchunk * 8;
        };
        auto set_b = [&](int tile) {
            const int n0 = tl.tile_n(tile) * BN;
#pragma unroll
            for (int i = 0; i < PB; ++i) b_off[i] = (n0 + 8 * (lw + 4 * i) + drow) * g.ldb + lchunk * 8;
        };
        set_a(atile);
        set_b(btile);
        auto issueA = [&](int stage) {
            unsigned char* st = smem + stage * STA + lw * 1024;
            const __hip_bfloat16* ap = g.A + ak * 64;
#pragma unroll
            for (int i = 0; i < PA; ++i)
                __builtin_amdgcn_global_load_lds((rs_gptr_t)(ap + a_off[i]), (rs_lptr_t)(st + i * 4096), 16, 0, 0);
            if (++ak == nk) {
                ak = 0;
                atile += tl.nslots;
                if (atile < tl.run1) set_a(atile);     // past the last tile: harmless re-loads keep vmcnt uniform
            }
        };
        auto issueB = [&](int stage) {
            unsigned char* st = smem + BOFF + stage * STB + lw * 1024;
            const __hip_bfloat16* bp = g.B + bk * 64;
#pragma unroll
            for (int i = 0; i < PB; ++i)
                __builtin_amdgcn_global_load_lds((rs_gptr_t)(bp + b_off[i]), (rs_lptr_t)(st + i * 4096), 16, 0, 0);
            if (++bk == nk) {
                bk = 0;
                btile += tl.nslots;
                if (btile < tl.run1) set_b(btile);
            }
        };
        // prologue: A(0) [A(1)], B(0)
        issueA(0);
        if (NSTA == 3) issueA(1);
        issueB(0);
        int fa = NSTA - 1, fb = 1;           // stages the next issues go to
        for (int step = 0; step < total; ++step) {
            // in flight may stay: nothing (NSTA = 2) or this wave's share of A(step + 1), issued after B(step)
            if (step == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTA == 3 ? PA : 0) : "memory");
            __builtin_amdgcn_s_barrier();    // B_step: the stages of `step` are complete, those of step - 1 free
            asm volatile("" ::: "memory");
            issueB(fb);                      // B(step + 1)
            issueA(fa);                      // A(step + NSTA - 1)
            fb ^= 1;
            fa = fa + 1 == NSTA ? 0 : fa + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land after the workgroup's LDS is released
        return;
    }

    // ---------------------------------------------------------------------- compute
    const int wm = wave / WNW, wn = wave % WNW;
    const int li = lane & 15, q = lane >> 4;
    const int swz = (li >> 1) & 7;
    // Software-pipelined K step, written instruction by instruction (the compiler placed every fragment read
    // right in front of the MFMAs that consume it, so each group of MFMAs waited out an LDS round trip, and both
    // waves of a SIMD did so in step with each other).  A step's 8 + 2 MB fragment reads form one list --
    // P0[0..3] (B rows, k 0..31), Q[0..MB) (A rows), P1[0..3] (k 32..63), Q'[0..MB) into the same slots -- and
    // its 2 MB groups of four MFMAs (one A fragment against the four B fragments) consume it in order.  After the
    // barrier a wave issues the first six reads of the NEW stage, then runs the LAST TWO groups of the previous step
    // from registers (8 MFMAs = 256 cycles of the SIMD's matrix pipe with its partner: the reads' round trip),
    // and from there on every group waits with a counted lgkmcnt for exactly its own fragments and issues the two
    // reads that are due 2-3 groups later.  lgkmcnt counts LDS operations in issue order; scalar loads, which
    // return out of order, can only make a counted wait longer, never shorter than needed.
    constexpr int G = 2 * MB, NRD = 8 + 2 * MB;
    static_assert(NB == 4 && MB >= 3, "pipelined loop: four B fragments per wave");
    const uint32_t lds0 = (uint32_t)(uintptr_t)(rs_lptr_t)smem;
    // this lane's fragment addresses in stage 0 (k chunk q; chunk 4 + q is the same address ^ 64), walked in place
    uint32_t a0 = lds0 + (wm * TM + li) * ROWB + ((q ^ swz) * 16);
    uint32_t b0 = lds0 + BOFF + (wn * TN + li) * ROWB + ((q ^ swz) * 16);
    rs_u32x4 P0[4], P1[4], Q[MB];
    rs_f32x4 acc[MB][NB];
    static_for<MB>([&](auto a) {
        static_for<NB>([&](auto b) { acc[a.v][b.v] = rs_f32x4{0.f, 0.f, 0.f, 0.f}; });
        Q[a.v] = rs_u32x4{0u, 0u, 0u, 0u};
    });
    static_for<4>([&](auto b) { P1[b.v] = rs_u32x4{0u, 0u, 0u, 0u}; });
    uint32_t a1, b1;
    auto rd = [&P0, &P1, &Q, &a0, &a1, &b0, &b1](auto r) {          // read number r of the step's list (captures named: a
        // generic lambda whose first instantiation leaves a branch out does not capture that branch's names by default)
        (void)P0; (void)P1; (void)Q; (void)a0; (void)a1; (void)b0; (void)b1;
        constexpr int R = decltype(r)::v;
        if constexpr (R < 4) RS_DSR(P0[R], b0, R * 16 * ROWB);
        else if constexpr (R < 4 + MB) RS_DSR(Q[R - 4], a0, (R - 4) * 16 * ROWB);
        else if constexpr (R < 8 + MB) RS_DSR(P1[R - 4 - MB], b1, (R - 4 - MB) * 16 * ROWB);
        else RS_DSR(Q[R - 8 - MB], a1, (R - 8 - MB) * 16 * ROWB);
    };
    auto group = [&P0, &P1, &Q, &acc](auto g) {       // the four MFMAs of group g
        (void)P0; (void)P1;
        constexpr int Gi = decltype(g)::v, A = Gi % MB;
        if constexpr (Gi < MB) {
            RS_MFMA(acc[A][0], P0[0], Q[A]); RS_MFMA(acc[A][1], P0[1], Q[A]);
            RS_MFMA(acc[A][2], P0[2], Q[A]); RS_MFMA(acc[A][3], P0[3], Q[A]);
        } else {
            RS_MFMA(acc[A][0], P1[0], Q[A]); RS_MFMA(acc[A][1], P1[1], Q[A]);
            RS_MFMA(acc[A][2], P1[2], Q[A]); RS_MFMA(acc[A][3], P1[3], Q[A]);
        }
    };
    auto epilogue = [&](int tile) {
        // the accumulators were written by MFMAs the compiler cannot see: give them their wait states before any
        // vector instruction reads them
        auto& ar = acc;
        static_for<MB>([&ar](auto a) {
            (void)ar;
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(ar[a.v][0]), "+v"(ar[a.v][1]), "+v"(ar[a.v][2]), "+v"(ar[a.v][3]));
        });
        // D[row = n][col = m]: lane = (m = li, q), register e -> n = 4 q + e.  (li, q re-derived here from an opaque
        // copy of the lane number: held across the K loop they cost registers the loop has none of to spare.)
        int le = lane;
        asm volatile("" : "+v"(le));
        const int li = le & 15, q = le >> 4;
        const int m0 = tl.tile_m(tile) * BM + wm * TM, n0 = tl.tile_n(tile) * BN + wn * TN;
#pragma unroll
        for (int a = 0; a < MB; ++a) {
            const int m = m0 + a * 16 + li;
            if constexpr (OUT == SP_F32) {
                float* crow = reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n0 + 4 * q;
                if (m < g.M) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) *reinterpret_cast<rs_f32x4*>(crow + b * 16) = acc[a][b];
                }
            } else {
                auto pk = [](float lo, float hi) {
                    if constexpr (IN == SP_F16)
                        return (uint32_t)__builtin_bit_cast(unsigned short, __float2half(lo)) |
                               ((uint32_t)__builtin_bit_cast(unsigned short, __float2half(hi)) << 16);
                    else
                        return (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(lo)) |
                               ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(hi)) << 16);
                };
                // rows (16-lane groups) 1 and 3 of x trade places with rows 0 and 2 of y: every lane then holds 8
                // consecutive n: columns 16 b + (q & 1) * 16 + (q >> 1) * 8 of its row
                rs_u32x4 o[NB / 2];
#pragma unroll
                for (int b = 0; b < NB; b += 2) {
                    const uint32_t x0 = pk(acc[a][b][0], acc[a][b][1]), x1 = pk(acc[a][b][2], acc[a][b][3]);
                    const uint32_t y0 = pk(acc[a][b + 1][0], acc[a][b + 1][1]), y1 = pk(acc[a][b + 1][2], acc[a][b + 1][3]);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                    o[b / 2] = rs_u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
                const int cq = (q & 1) * 16 + (q >> 1) * 8;
                __hip_bfloat16* crow = reinterpret_cast<__hip_bfloat16*>(g.C) + (int64_t)m * g.ldc + n0 + cq;
#pragma unroll
                for (int b = 0; b < NB; b += 2) {
                    if (m < g.M) *reinterpret_cast<rs_u32x4*>(crow + b * 16) = o[b / 2];
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] = rs_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ... and the fresh zeros theirs before the next MFMA takes them as its addend
        static_for<MB>([&ar](auto a) {
            (void)ar;
            asm volatile("s_nop 7" : "+v"(ar[a.v][0]), "+v"(ar[a.v][1]), "+v"(ar[a.v][2]), "+v"(ar[a.v][3]));
        });
    };

    int ctile = tl.first, ck = 0, stageA = 0, stageB = 0;
    for (int step = 0;; ++step) {
        const bool more = step < total;      // the pass after the last step only drains the deferred groups
        if (more) {
            a1 = a0 ^ 64u;
            b1 = b0 ^ 64u;
            __builtin_amdgcn_s_barrier();    // B_step: this step's stages have landed, the previous step's are free
            asm volatile("" ::: "memory");
            static_for<6>([&](auto r) { rd(r); });
        }
        group(IC<G - 2>{});                  // the previous step's last two groups (zeros on the first pass)
        group(IC<G - 1>{});
        if (ck == nk) {                      // ... which completed a tile
            epilogue(ctile);
            ck = 0;
            ctile += tl.nslots;
        }
        if (!more) break;
        static_for<G - 2>([&](auto gi) {
            constexpr int Gi = decltype(gi)::v;
            constexpr int issued = (6 + 2 * Gi < NRD) ? 6 + 2 * Gi : NRD;
            constexpr int need = Gi < MB ? 4 + Gi : 8 + Gi;
            RS_WAIT(issued - need - 1);
            if constexpr (6 + 2 * Gi < NRD) rd(IC<6 + 2 * Gi>{});
            if constexpr (7 + 2 * Gi < NRD) rd(IC<7 + 2 * Gi>{});
            group(gi);
        });
        RS_WAIT(0);                          // every read of this stage has returned before the barrier frees it
        a0 += stageA + 1 == NSTA ? -(NSTA - 1) * STA : STA;
        b0 += stageB + 1 == NSTB ? -(NSTB - 1) * STB : STB;
        stageA = stageA + 1 == NSTA ? 0 : stageA + 1;
        stageB = stageB + 1 == NSTB ? 0 : stageB + 1;
        ++ck;
    }
}

template <int BM, int BN, int WMW, int NSTA>
static int launch_nt_rs(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                        int in_dtype, int out_dtype, hipStream_t s) {
    RsArgs g;
    g.A = (const __hip_bfloat16*)A; g.B = (const __hip_bfloat16*)B; g.C = C;
    g.M = M; g.N = N; g.K = K; g.lda = (int)lda; g.ldb = (int)ldb; g.ldc = ldc;
    g.ntn = N / BN; g.ntm = (M + BM - 1) / BM;
    // XCD grid gcols x (8 / gcols): every XCD streams its rows of A once and its columns of B once per round,
    // so the traffic from beyond the L2s is about gcols * |A| + (8 / gcols) * |B|; take the cheapest split.
    // (The data gradient, whose A = gU is 24x its B, wants gcols = 1: measured 170 -> 148 us.)
    g.gcols = 1;
    double best = -1;
    for (int c = 1; c <= 8; c <<= 1) {
        if (g.ntn % c) continue;
        const double cost = (double)c * M * K + (8.0 / c) * N * K;
        if (best < 0 || cost < best) best = cost, g.gcols = c;
    }
    // Within an XCD the 32 workgroups run 32 tiles at a time in step with one another, and what the L2 saves is the
    // sharing among THOSE: a footprint of 4 columns x 8 rows shares every A line 4 ways and every B line 8 ways
    // (12 columns x 2.7 rows: 12 and 2.7 -- the wider B panels then miss 38 % of the time).  Measured: forward
    // 144.5 -> 139.9 us, layer 0 228 -> 216 us; column groups of 2, 3 or 6 are no better than the full width.
    g.cgrp = g.ntn / g.gcols;
    if (g.cgrp % 4 == 0) g.cgrp = 4;
#define SP_RS_LAUNCH(OUT, IN) hipLaunchKernelGGL((gemm_nt_rs_kernel<BM, BN, WMW, NSTA, OUT, IN>), dim3(256), dim3(768), 0, s, g)
    if (in_dtype == SP_F16) {
        if (out_dtype == SP_F32) SP_RS_LAUNCH(SP_F32, SP_F16); else SP_RS_LAUNCH(SP_F16, SP_F16);
    } else {
        if (out_dtype == SP_F32) SP_RS_LAUNCH(SP_F32, SP_BF16); else SP_RS_LAUNCH(SP_BF16, SP_BF16);
    }
#undef SP_RS_LAUNCH
    return check_launch("gemm_nt_rs_kernel");
}

// shape rules shared by the variants; the caller has validated pointers and dtypes
bool nt_rs_ok(int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, const void* A, const void* B, const void* C,
              int out_dtype, int BN) {
    return N % BN == 0 && K % 64 == 0 && K >= 64 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N &&
           ldc % (out_dtype == SP_F32 ? 4 : 8) == 0 && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0 &&
           (int64_t)M * lda < ((int64_t)1 << 30) && (int64_t)N * ldb < ((int64_t)1 << 30) && M >= 1;
}

int gemm_nt_rs(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
               int in_dtype, int out_dtype, int variant, hipStream_t s) {
    switch (variant) {
        case 0: return launch_nt_rs<192, 256, 2, 3>(A, B, C, M, N, K, lda, ldb, ldc, in_dtype, out_dtype, s);     // N % 256 == 0
        case 3: return launch_nt_rs<256, 128, 4, 3>(A, B, C, M, N, K, lda, ldb, ldc, in_dtype, out_dtype, s);     // N % 128 == 0
        default: return fail(SP_EINVAL, "gemm_nt_rs: variant %d", variant);
    }
}


// -----------------------------------------------------------------------------------------------------
//   TN:  P[z][i][j] = sum_{k in range z} X[k][i] * G[k][j]    (both operands stored k-major, i.e. row-
//        contiguous in i / j: the weight gradients dW = LN(x)^T gU of the SRU layers and g^T x of the
//        classifier, K = T'*B rows split into S ranges whose f32 partial products go to slabs)
// Same roles and ring as the NT kernel; what differs is the operand path:
//   * a stage holds 64 k-rows of a 256-column X panel and a 256-column G panel (512-byte rows); an LDS-DMA
//     instruction moves two whole rows.  The loaders use BUFFER loads whose descriptor ends after the last
//     valid k-row: rows past K (the last, partial K step) read as zeros, so no tail pass is needed;
//   * the MFMA wants 8 consecutive k per lane, the rows hold consecutive i / j: fragments are read with
//     ds_read_b64_tr_b16, which hands each lane one COLUMN of a 4 (k) x 16 (i) block.  A 32-lane half reads
//     rows k..k+3 and k+8..k+11 of one 32-byte column group; 32-byte group c of row k is stored at group
//     c ^ ((k & 3) | ((k >> 3) & 1) << 2), which spreads those 8 rows over the 8 groups of a 256-byte
//     bank row: conflict-free (again applied on the DMA's source address).
struct TnArgs {
    const __hip_bfloat16* X; const __hip_bfloat16* G; float* P;
    int Mi, Nj, K, ldx, ldg;
    int64_t ldp, slab;                // row pitch and slab stride of P, in floats
    int nti, ntj, S, steps_per;       // tiles, splits, K steps (of 64) per split
};

typedef short rs_s16x4 __attribute__((ext_vector_type(4)));
struct TnCursor { int item, step, steps, soff; };

__global__ __launch_bounds__(768, 3) void gemm_tn_rs_kernel(TnArgs g) {
    constexpr int BI = 256, BJ = 256;
    constexpr int NSX = 2, NSG = 3;               // ring depths: X panels two stages, G panels (gU: the larger, colder operand) three
    constexpr int MB = 8, NB = 4;                 // wave tile 128 (i) x 64 (j), waves 2 (i) x 4 (j)
    constexpr int PART = 64 * 512, GOFF = NSX * PART;
    constexpr int PER = 8;                        // DMA instructions per loader wave, stage and operand
    __shared__ __attribute__((aligned(1024))) unsigned char smem[(NSX + NSG) * PART];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // items = (split z, tile): an XCD takes a contiguous run of the z-major list, so its workgroups share
    // the X and G panels of (mostly) one K range in their L2
    const int ntiles = g.nti * g.ntj, nitems = ntiles * g.S;
    const int per_xcd = (nitems + 7) / 8, nslots = gridDim.x >> 3;
    const int x_lo = (blockIdx.x & 7) * per_xcd, x_hi = min(x_lo + per_xcd, nitems);
    const int first = x_lo + (blockIdx.x >> 3);
    if (first >= x_hi) return;
    const int count = (x_hi - first + nslots - 1) / nslots;
    const int nsteps_all = (g.K + 63) / 64;
    auto item_steps = [&](int item) {
        const int z = item / ntiles;
        return max(min(nsteps_all - z * g.steps_per, g.steps_per), 0);
    };
    int total = 0;
    for (int c = 0; c < count; ++c) total += item_steps(first + c * nslots);

    if (wave >= 8) {
        // ------------------------------------------------------------------ loader
        const int lw = wave - 8;
        const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)g.X, 0, (int)((int64_t)g.K * g.ldx * 2), 0x00020000);
        const auto rg = __builtin_amdgcn_make_buffer_rsrc((void*)g.G, 0, (int)((int64_t)g.K * g.ldg * 2), 0x00020000);
        int vx[8], vg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = 2 * (lw + 4 * i) + (lane >> 5);          // row within the stage (both parts)
            const int sk = (k & 3) | (((k >> 3) & 1) << 2);
            const int c = (lane & 31) ^ (sk << 1);
            vx[i] = k * g.ldx * 2 + c * 16;
            vg[i] = k * g.ldg * 2 + c * 16;
        }
        // one cursor per operand over the flat (item, k step) sequence: G runs one step further ahead than X
        auto set_cur = [&](TnCursor& c, int item, bool isG) {
            c.item = item; c.step = 0;
            if (item < x_hi) {
                const int z = item / ntiles, t = item % ntiles;
                const int k0 = z * g.steps_per * 64;
                c.soff = isG ? (k0 * g.ldg + (t / g.nti) * BJ) * 2 : (k0 * g.ldx + (t % g.nti) * BI) * 2;
                c.steps = item_steps(item);
            } else {
                c.steps = 1 << 30;      // past the last item: keeps reading the rows that follow (zeros beyond the buffer)
            }
        };
        TnCursor cx, cg;
        set_cur(cx, first, false);
        set_cur(cg, first, true);
        auto issueX = [&](int stage) {
            unsigned char* st = smem + stage * PART + lw * 1024;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (rs_lptr_t)(st + i * 4096), 16, vx[i], cx.soff, 0, 0);
            cx.soff += 64 * g.ldx * 2;
            if (++cx.step == cx.steps) set_cur(cx, cx.item + nslots, false);
        };
        auto issueG = [&](int stage) {
            unsigned char* st = smem + GOFF + stage * PART + lw * 1024;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rg, (rs_lptr_t)(st + i * 4096), 16, vg[i], cg.soff, 0, 0);
            cg.soff += 64 * g.ldg * 2;
            if (++cg.step == cg.steps) set_cur(cg, cg.item + nslots, true);
        };
        issueG(0);
        issueG(1);
        issueX(0);
        int fx = 1, fg = 2;
        for (int step = 0; step < total; ++step) {
            // in flight may stay: this wave's share of G(step + 1), issued after X(step)
            if (step == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issueX(fx);                      // X(step + 1)
            issueG(fg);                      // G(step + 2)
            fx ^= 1;
            fg = fg + 1 == NSG ? 0 : fg + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---------------------------------------------------------------------- compute
    const int wi = wave >> 2, wj = wave & 3;
    const int li = lane & 15, q = lane >> 4, qq = li >> 2, p = li & 3;
    // transposed-read address of this lane inside a [64 k][256] part: row 8 q + qq (+ 32 kk, + 4 for the second
    // half of the fragment), 32-byte group (block) ^ sk, byte (p >> 1) * 16 + (p & 1) * 8 inside the group.
    // block = 8 wi + a (X) / 4 wj + b (G) with sk < 8 and a < 8, b < 4: fragment a's address is fragment 0's ^ (a << 5)
    // (one v_xor per fragment instead of twelve address registers: the registers go to the fragment pipeline below)
    const int sk = qq | ((q & 1) << 2);
    const int rowoff = (8 * q + qq) * 512 + (p >> 1) * 16 + (p & 1) * 8;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(rs_lptr_t)smem;
    const uint32_t xb0 = lds0 + rowoff + (((wi * 8) ^ sk) << 5);
    const uint32_t gb0 = lds0 + GOFF + rowoff + ((((wj * 4) & 15) ^ sk) << 5);

    rs_f32x4 acc[MB][NB];
#pragma unroll
    for (int a = 0; a < MB; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = rs_f32x4{0.f, 0.f, 0.f, 0.f};

    auto frag = [&](uint32_t addr) {
        typedef __attribute__((address_space(3))) rs_s16x4* lp;
        const rs_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)addr);
        const rs_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(addr + 4 * 512));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(rs_bf16x8, v);
    };

    int citem = first, ck = 0, csteps = item_steps(first), sx = 0, sg = 0;
    for (int step = 0; step < total; ++step) {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // Fragment pipeline: the X fragment of group n + 1 (a group = one X fragment against the wave's four G
        // fragments) is requested BEFORE the four MFMAs of group n, the G set of the step's second half fragment by
        // fragment behind the last MFMAs that read the first half's; the sched_group_barrier sequence pins that order
        // (left alone, the scheduler sinks every read to just in front of its use: two reads, a full wait and four
        // MFMAs per group, the LDS round trip exposed sixteen times a step).
        const uint32_t xb = xb0 + sx * PART, gb = gb0 + sg * PART;
        rs_bf16x8 Pf[NB], Q, Qn;
#pragma unroll
        for (int b = 0; b < NB; ++b) Pf[b] = frag(gb ^ (uint32_t)(b << 5));
        Q = frag(xb);
        __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int kk = n >> 3, a = n & 7;
            if (n < 15) {
                Qn = frag((xb ^ (uint32_t)(((n + 1) & 7) << 5)) + ((n + 1) >> 3) * 32 * 512);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Pf[b], Q, acc[a][b], 0, 0, 0);
                if (n == 7) {     // the second half's G fragment b, behind the last MFMA that reads the first half's
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    Pf[b] = frag((gb ^ (uint32_t)(b << 5)) + 32 * 512);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
            }
            if (n != 7) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            (void)kk;
            Q = Qn;
        }
        asm volatile("" ::: "memory");
        sx ^= 1;
        sg = sg + 1 == NSG ? 0 : sg + 1;
        if (++ck == csteps) {
            // D[row = j][col = i]: lane = (i = li, q), register e -> j = 4 q + e (li, q re-derived from an opaque copy of
            // the lane number: held across the K loop they cost registers the loop has none of to spare)
            int le = lane;
            asm volatile("" : "+v"(le));
            const int eli = le & 15, eq = le >> 4;
            const int z = citem / ntiles, t = citem % ntiles;
            const int i0 = (t % g.nti) * BI + wi * 128, j0 = (t / g.nti) * BJ + wj * 64;
            float* Pz = g.P + (int64_t)z * g.slab;
#pragma unroll
            for (int a = 0; a < MB; ++a) {
                float* crow = Pz + (int64_t)(i0 + a * 16 + eli) * g.ldp + j0 + 4 * eq;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    *reinterpret_cast<rs_f32x4*>(crow + b * 16) = acc[a][b];
                    acc[a][b] = rs_f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            ck = 0;
            citem += nslots;
            csteps = citem < x_hi ? item_steps(citem) : (1 << 30);
        }
    }
}

bool tn_rs_ok(int Mi, int Nj, int K, int64_t ldx, int64_t ldg, const void* X, const void* G, const void* P) {
    return Mi % 256 == 0 && Nj % 256 == 0 && K >= 64 && ldx % 8 == 0 && ldg % 8 == 0 && ldx >= Mi && ldg >= Nj &&
           (((uintptr_t)X | (uintptr_t)G | (uintptr_t)P) & 15) == 0 && (int64_t)K * ldx < ((int64_t)1 << 30) &&
           (int64_t)K * ldg < ((int64_t)1 << 30);
}

// S = number of K ranges: the smallest that gives every CU an item (at most 64, at least 8 K steps each)
int tn_rs_splits(int Mi, int Nj, int K) {
    const int tiles = (Mi / 256) * (Nj / 256), nsteps = (K + 63) / 64;
    int S = 256 / tiles;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    while (S > 1 && (nsteps / S < 8 || (S - 1) * ((nsteps + S - 1) / S) >= nsteps)) --S;   // no empty range
    return S;
}

int gemm_tn_rs(const void* X, const void* G, float* P, int Mi, int Nj, int K, int64_t ldx, int64_t ldg, int64_t ldp,
               int64_t slab_stride, int S, int variant, hipStream_t s) {
    TnArgs g;
    g.X = (const __hip_bfloat16*)X; g.G = (const __hip_bfloat16*)G; g.P = P;
    g.Mi = Mi; g.Nj = Nj; g.K = K; g.ldx = (int)ldx; g.ldg = (int)ldg;
    g.ldp = ldp; g.slab = slab_stride;
    g.nti = Mi / 256; g.ntj = Nj / 256; g.S = S;
    g.steps_per = ((K + 63) / 64 + S - 1) / S;
    (void)variant;
    hipLaunchKernelGGL(gemm_tn_rs_kernel, dim3(256), dim3(768), 0, s, g);
    return check_launch("gemm_tn_rs_kernel");
}

}  // namespace sp
