// bf16 GEMM  C[M][N] = A[M][K] * B[N][K]^T  (both operands K-contiguous, f32 accumulate) for the SRU
// projection layers of the encoder (net.py:61-72: sru.SRU's weight matmul; U = LN(x) W, and its data
// gradient gX = gU W^T -- with W kept in both layouts by sp_sru_pack_w the two are the same "NT"
// problem).  Hand-written for gfx950:
//   * workgroup tile 256 (M) x 256 or 128 (N), K step 32, 8 or 4 waves, each 128 x 64 of output in 8
//     accumulators of v_mfma_f32_32x32x16_bf16 (128 VGPRs);
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) through a
//     three- or four-stage ring with a COUNTED vmcnt, one s_barrier per K step;
//   * LDS rows are 64 bytes (4 chunks); the chunk a lane DMA-writes is XOR-ed with (row >> 2) & 3 so
//     that the 16 lanes a ds_read_b128 serves per cycle hit 16 different bank groups;
//   * the MFMA takes the B tile as its row operand and the A tile as its column operand, so a lane
//     ends up with 4 consecutive n of one m: 8-byte stores straight into C, no LDS transpose.
// Tile order: N fastest inside one XCD's run of tiles, so the workgroups that share an A row panel
// share an L2.
#include <type_traits>

#include "gemm.h"

namespace sp {

typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef float g_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t g_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t g_u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(1))) const void* g_gptr_t;
typedef __attribute__((address_space(3))) void* g_lptr_t;

constexpr int kBM = 256, kBK = 32;
constexpr int kRowB = kBK * 2;                       // 64 bytes per LDS row

// WN = wave columns (2 or 4): tile 256 x (64 WN), 2 WN waves, each wave 128 x 64 of output.
template <int WN>
__global__ __launch_bounds__(128 * WN, WN == 2 ? 2 : 1) void gemm_nt_bf16_kernel(
    const __hip_bfloat16* __restrict__ A, const __hip_bfloat16* __restrict__ B,
    __hip_bfloat16* __restrict__ C, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc) {
    constexpr int BN = 64 * WN, NW = 2 * WN;
    constexpr int NS = WN == 2 ? 3 : 4;                  // ring depth: 72 KB (2 workgroups / CU) or 128 KB
    constexpr int STAGE = (kBM + BN) * kRowB;
    constexpr int GA = (kBM / 16) / NW, GB = (BN / 16) / NW;   // 16-row DMA groups per wave and stage
    constexpr int PER_STAGE = GA + GB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = N / BN, ntm = (M + kBM - 1) / kBM;
    // XCD-aware order: workgroups are dealt round-robin to the 8 XCDs, so give every XCD a contiguous
    // run of tiles (N fastest): the workgroups that share an A row panel then share one L2
    const int ntiles = ntn * ntm;
    const int per_xcd = (ntiles + 7) / 8;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int tn = tile % ntn, tm = tile / ntn;
    const int m0 = tm * kBM, n0 = tn * BN;
    const int wm = wave / WN, wn = wave % WN;          // this wave's 128 x 64 sub-tile
    const int nk = K / kBK;

    // ---- DMA addressing: one instruction = 16 rows x 4 chunks; lane = (row in group, physical chunk)
    const int drow = lane >> 2, dchunk = lane & 3;
    const int lchunk = dchunk ^ ((drow >> 2) & 3);    // logical (global) chunk stored at this position
    const __hip_bfloat16* a_src[GA];
    const __hip_bfloat16* b_src[GB];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int row = min(m0 + (wave * GA + i) * 16 + drow, M - 1);   // clamp: rows past M are never stored
        a_src[i] = A + (int64_t)row * lda + lchunk * 8;
    }
#pragma unroll
    for (int i = 0; i < GB; ++i) {
        const int row = n0 + (wave * GB + i) * 16 + drow;
        b_src[i] = B + (int64_t)row * ldb + lchunk * 8;
    }
    auto issue = [&](int kt) {
        unsigned char* st = smem + (kt % NS) * STAGE;
        const int k0 = min(kt, nk - 1) * kBK;         // past the end: harmless re-loads keep vmcnt uniform
#pragma unroll
        for (int i = 0; i < GA; ++i)
            __builtin_amdgcn_global_load_lds((g_gptr_t)(a_src[i] + k0), (g_lptr_t)(st + (wave * GA + i) * 16 * kRowB), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < GB; ++i)
            __builtin_amdgcn_global_load_lds((g_gptr_t)(b_src[i] + k0),
                                             (g_lptr_t)(st + kBM * kRowB + (wave * GB + i) * 16 * kRowB), 16, 0, 0);
    };

    g_f32x16 acc[2][4];   // [n block of 32][m block of 32]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;

    // fragment addressing: lane (r = row within a 32-block, h = k half); logical chunk kk*2 + h
    const int r = lane & 31, h = lane >> 5;
    const int swz = (r >> 2) & 3;                     // 32-blocks start at multiples of 32 rows
#pragma unroll
    for (int s0 = 0; s0 < NS - 1; ++s0) issue(s0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER_STAGE) : "memory");   // this wave's share of stage kt landed
        __builtin_amdgcn_s_barrier();                        // ... and everybody's; stage kt-1 is free
        asm volatile("" ::: "memory");
        issue(kt + NS - 1);
        const unsigned char* st = smem + (kt % NS) * STAGE;
        const unsigned char* sa = st + (wm * 128 + r) * kRowB;
        const unsigned char* sb = st + kBM * kRowB + (wn * 64 + r) * kRowB;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int pc = ((kk * 2 + h) ^ swz) * 16;
            g_bf16x8 fb[2], fa[4];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                fb[a] = __builtin_bit_cast(g_bf16x8, *reinterpret_cast<const uint4*>(sb + a * 32 * kRowB + pc));
#pragma unroll
            for (int b = 0; b < 4; ++b)
                fa[b] = __builtin_bit_cast(g_bf16x8, *reinterpret_cast<const uint4*>(sa + b * 32 * kRowB + pc));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[a], fa[b], acc[a][b], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the clamped tail loads before the LDS is released

    // ---- epilogue: D[row = n][col = m]: lane = m (r), registers q -> n = 8 (q >> 2) + 4 h + (q & 3)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int m = m0 + wm * 128 + b * 32 + r;
        if (m < M) {
            __hip_bfloat16* crow = C + (int64_t)m * ldc + n0 + wn * 64;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* v = reinterpret_cast<const float*>(&acc[a][b]) + 4 * g;
                    uint2 o;
                    o.x = (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[0])) |
                          ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[1])) << 16);
                    o.y = (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[2])) |
                          ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[3])) << 16);
                    *reinterpret_cast<uint2*>(crow + a * 32 + 8 * g + 4 * h) = o;
                }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Persistent variant for the big projection shapes (N % 256 == 0, K % 64 == 0): one workgroup per CU
// walks its own list of (64 MB) x 256 tiles as ONE flat sequence of K steps, so nothing drains between
// tiles:
//   * K step 64: every LDS row is one whole 128-byte line of the operand (a DMA instruction fetches 8
//     rows x 128 B), two stages; the stage for step g+2 is requested as soon as step g's last fragments
//     have been read, also across a tile boundary (the load cursor runs ahead of the compute cursor);
//   * inside a step the fragments of sub-step j+1 are read while sub-step j's MFMAs issue, and the one
//     barrier of the step sits between sub-steps 2 and 3 with the MFMA pipe still full;
//   * the epilogue goes through a wave-private 4 KB LDS patch (no barrier) so that every store
//     instruction writes 8 whole 128-byte lines of C, and drains under the next tile's K steps;
//   * chunk c of row r lives at chunk c ^ ((r >> 1) & 7): conflict-free for the ds_read_b128 lane groups.
// Tile height 256 or 192 is chosen on the host to minimise the rounds x height product.
template <int MB>
__global__ __launch_bounds__(512, 1) void gemm_nt_persist_kernel(
    const __hip_bfloat16* __restrict__ A, const __hip_bfloat16* __restrict__ B,
    __hip_bfloat16* __restrict__ C, int M, int N, int K, int lda, int ldb, int64_t ldc, int ntn, int ntm, int gcols) {
    constexpr int BM = 64 * MB, WM = 32 * MB, ROWB = 128;
    constexpr int STAGE = (BM + 256) * ROWB, EPI = 4096;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + 8 * EPI];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int nk = K / 64;
    // XCD-aware tile ownership.  The eight XCDs (workgroup b runs on XCD b % 8) form a gcols x (8 / gcols)
    // grid over the tile matrix: an XCD owns ntn / gcols whole tile COLUMNS of a contiguous range of tile
    // rows, so its share of B (at most ~2 MB, chosen on the host) stays resident in its 4 MB L2 and every
    // A panel it streams is used by all of its columns back to back.  Its workgroups take the tiles of
    // that sub-matrix round-robin, columns fastest.
    const int xcd = blockIdx.x & 7, nslots = gridDim.x >> 3;
    const int cols_per = ntn / gcols, grows = 8 / gcols;
    const int col0 = (xcd % gcols) * cols_per;
    const int row_lo = (int)((int64_t)(xcd / gcols) * ntm / grows), row_hi = (int)((int64_t)(xcd / gcols + 1) * ntm / grows);
    const int run1 = (row_hi - row_lo) * cols_per;       // tiles this XCD owns, local index 0 .. run1-1
    const int first = blockIdx.x >> 3;
    if (first >= run1) return;
    const int count = (run1 - first + nslots - 1) / nslots;
    auto tile_m = [&](int i) { return row_lo + i / cols_per; };
    auto tile_n = [&](int i) { return col0 + i % cols_per; };
    const int total = count * nk;

    // ---- load cursor
    const int drow = lane >> 3;
    const int lchunk = (lane & 7) ^ (((wave & 1) << 2) | (drow >> 1));   // row = (wave + 8 i) * 8 + drow
    int ltile = first, lk = 0;
    int a_off[MB], b_off;
    auto set_tile = [&](int tile) {
        const int m0 = tile_m(tile) * BM, n0 = tile_n(tile) * 256;
#pragma unroll
        for (int i = 0; i < MB; ++i)
            a_off[i] = min(m0 + (wave + 8 * i) * 8 + drow, M - 1) * lda + lchunk * 8;   // rows past M are never stored
        b_off = (n0 + wave * 8 + drow) * ldb + lchunk * 8;
    };
    set_tile(first);
    auto issue = [&](int s) {
        unsigned char* st = smem + s * STAGE + wave * 1024;
        const __hip_bfloat16* ap = A + lk * 64;
        const __hip_bfloat16* bp = B + lk * 64 + b_off;
#pragma unroll
        for (int i = 0; i < MB; ++i)
            __builtin_amdgcn_global_load_lds((g_gptr_t)(ap + a_off[i]), (g_lptr_t)(st + i * 8192), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((g_gptr_t)(bp + (int64_t)i * 64 * ldb), (g_lptr_t)(st + BM * ROWB + i * 8192), 16, 0, 0);
        if (++lk == nk) {
            lk = 0;
            ltile += nslots;
            if (ltile < run1) set_tile(ltile);
        }
    };

    // ---- fragments: lane (r = row in a 32-block, h = k half of a 16-wide sub-step)
    const int r = lane & 31, h = lane >> 5;
    int pc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pc[j] = ((j * 2 + h) ^ ((r >> 1) & 7)) * 16;
    const int fa_base = (wm * WM + r) * ROWB, fb_base = BM * ROWB + (wn * 64 + r) * ROWB;
    // The fragment reads and their waits are written out by hand: left to the compiler, a ds_read that
    // follows an LDS-DMA into the same array is preceded by s_waitcnt vmcnt(0) (it cannot see that the
    // DMA fills the OTHER stage), which would serialise every load behind the step that issued it.
    // All asm statements here are volatile (kept in program order) and the waits name the fragments
    // as in-out operands, so no MFMA can be scheduled above the wait that covers its operands.
    g_u32x4 fa[2][MB], fb[2][2];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(g_lptr_t)smem;
    const uint32_t fa_addr = lds0 + fa_base, fb_addr = lds0 + fb_base;
    auto read_frags = [&](int buf, int s, int j) {
        const uint32_t off = (uint32_t)(s * STAGE) + (uint32_t)pc[j];
        const uint32_t pa = fa_addr + off, pb = fb_addr + off;
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[buf][0]) : "v"(pb) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fb[buf][1]) : "v"(pb) : "memory");
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[buf][0]) : "v"(pa) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(fa[buf][1]) : "v"(pa) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(fa[buf][2]) : "v"(pa) : "memory");
        if constexpr (MB == 4) asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(fa[buf][3]) : "v"(pa) : "memory");
    };
    // wait until at most `n` LDS operations are outstanding; covers the fragments of buffer `buf`
#define SP_WAIT_FRAGS(buf, n)                                                                                          \
    do {                                                                                                               \
        if constexpr (MB == 4)                                                                                         \
            asm volatile("s_waitcnt lgkmcnt(%6)"                                                                       \
                         : "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fa[buf][2]),   \
                           "+v"(fa[buf][MB - 1])                                                                       \
                         : "n"(n)                                                                                      \
                         : "memory");                                                                                  \
        else                                                                                                           \
            asm volatile("s_waitcnt lgkmcnt(%5)"                                                                       \
                         : "+v"(fb[buf][0]), "+v"(fb[buf][1]), "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fa[buf][2])    \
                         : "n"(n)                                                                                      \
                         : "memory");                                                                                  \
    } while (0)
    constexpr int NR = MB + 2;     // reads per sub-step
    g_f32x16 acc[2][MB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < MB; ++b)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    };
    auto mfma = [&](int buf) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < MB; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, fb[buf][a]),
                                                                    __builtin_bit_cast(g_bf16x8, fa[buf][b]), acc[a][b], 0, 0, 0);
    };
    // D[row = n][col = m]: lane = m (r), register q -> n = 8 (q >> 2) + 4 h + (q & 3)
    auto epilogue = [&](int tile) {
        const int m0 = tile_m(tile) * BM + wm * WM, n0 = tile_n(tile) * 256 + wn * 64;
        const int erow = lane >> 3, ephys = lane & 7;
        const uint32_t ep = lds0 + 2 * STAGE + wave * EPI;
        const uint32_t ep_w = ep + r * ROWB + h * 8, ep_r = ep + erow * ROWB + ephys * 16;
#pragma unroll
        for (int b = 0; b < MB; ++b) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float* v = reinterpret_cast<const float*>(&acc[a][b]) + 4 * g;
                    g_u32x2 o;
                    o.x = (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[0])) |
                          ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[1])) << 16);
                    o.y = (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[2])) |
                          ((uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(v[3])) << 16);
                    const uint32_t wa = ep_w + (((a * 4 + g) ^ (r & 7)) * 16);
                    asm volatile("ds_write_b64 %0, %1" ::"v"(wa), "v"(o) : "memory");
                }
            g_u32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(ep_r + i * 8 * ROWB) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + erow;
                const int m = m0 + b * 32 + row;
                if (m < M) *reinterpret_cast<g_u32x4*>(C + (int64_t)m * ldc + n0 + (ephys ^ erow) * 8) = v[i];
            }
        }
    };

    zero_acc();
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (total > 1) issue(1);
    read_frags(0, 0, 0);
    int ctile = first, ck = 0;
    for (int step = 0; step < total; ++step) {
        const int s = step & 1;
        read_frags(1, s, 1);
        SP_WAIT_FRAGS(0, NR);
        mfma(0);
        read_frags(0, s, 2);
        SP_WAIT_FRAGS(1, NR);
        mfma(1);
        read_frags(1, s, 3);
        SP_WAIT_FRAGS(0, NR);
        mfma(0);
        // this wave's share of stage s^1 (step + 1) has landed and its reads of stage s are complete
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SP_WAIT_FRAGS(1, 0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (step + 2 < total) issue(s);
        read_frags(0, s ^ 1, 0);         // (after the very last step this reads a stale stage: harmless)
        mfma(1);
        if (++ck == nk) {
            epilogue(ctile);
            zero_acc();
            ck = 0;
            ctile += nslots;
        }
    }
    SP_WAIT_FRAGS(0, 0);
#undef SP_WAIT_FRAGS
}


}  // namespace sp

extern "C" int sp_gemm_nt_bf16(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda,
                               int64_t ldb, int64_t ldc, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(M >= 0 && N >= 1 && K >= 1, "sp_gemm_nt_bf16: bad shape M=%d N=%d K=%d", M, N, K);
    SP_CHECK_ARG(N % 128 == 0 && K % kBK == 0, "sp_gemm_nt_bf16: N must be a multiple of 128 and K of %d", kBK);
    SP_CHECK_ARG(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && lda >= K && ldb >= K && ldc >= N,
                 "sp_gemm_nt_bf16: leading dimensions must cover the rows and keep 16-byte alignment");
    if (M == 0) return SP_OK;
    SP_CHECK_ARG(A && B && C, "sp_gemm_nt_bf16: null pointer");
    SP_CHECK_ARG((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && ((uintptr_t)C & 7) == 0,
                 "sp_gemm_nt_bf16: operands must be 16-byte aligned");
    if (N % 256 == 0 && K % 64 == 0 && ldc % 8 == 0 && ((uintptr_t)C & 15) == 0 && (int64_t)M * lda < ((int64_t)1 << 30) &&
        (int64_t)N * ldb < ((int64_t)1 << 30) && (int64_t)((M + 191) / 192) * (N / 256) >= 256) {
        // persistent kernel: tile height 256 or 192, whichever needs fewer row-rounds on the 256 CUs
        int best_mb = 4;
        int64_t best_cost = -1;
        for (int mb = 4; mb >= 3; --mb) {
            const int64_t t = (int64_t)((M + 64 * mb - 1) / (64 * mb)) * (N / 256);
            const int64_t rounds = ((t + 7) / 8 + 31) / 32;
            if (best_cost < 0 || rounds * mb < best_cost) best_cost = rounds * mb, best_mb = mb;
        }
        const int ntn = N / 256, ntm = (M + 64 * best_mb - 1) / (64 * best_mb);
        SP_CHECK_ARG((int64_t)ntm * ntn < ((int64_t)1 << 30), "sp_gemm_nt_bf16: too many tiles");
        // columns per XCD: as few as divide evenly, so that an XCD's B panels (256 x K x 2 bytes each) fit in
        // half of its 4 MB L2
        int gcols = 1;
        for (int g = 8; g >= 1; g >>= 1)
            if (ntn % g == 0 && (int64_t)(ntn / g) * 256 * K * 2 <= (2 << 20)) gcols = g;
        hipStream_t ps = reinterpret_cast<hipStream_t>(stream);
        if (best_mb == 4)
            hipLaunchKernelGGL((gemm_nt_persist_kernel<4>), dim3(256), dim3(512), 0, ps, (const __hip_bfloat16*)A,
                               (const __hip_bfloat16*)B, (__hip_bfloat16*)C, M, N, K, (int)lda, (int)ldb, ldc, ntn, ntm, gcols);
        else
            hipLaunchKernelGGL((gemm_nt_persist_kernel<3>), dim3(256), dim3(512), 0, ps, (const __hip_bfloat16*)A,
                               (const __hip_bfloat16*)B, (__hip_bfloat16*)C, M, N, K, (int)lda, (int)ldb, ldc, ntn, ntm, gcols);
        return check_launch("gemm_nt_persist_kernel");
    }
    const bool wide = N % 256 == 0 && (int64_t)((M + kBM - 1) / kBM) * (N / 256) >= 256;   // enough 256 x 256 tiles to fill the chip
    const int bn = wide ? 256 : 128;
    const int64_t tiles = (int64_t)((M + kBM - 1) / kBM) * (N / bn);
    SP_CHECK_ARG(tiles < ((int64_t)1 << 30), "sp_gemm_nt_bf16: too many tiles");
    const unsigned grid = (unsigned)((tiles + 7) / 8 * 8);   // whole rounds over the 8 XCDs (surplus workgroups exit)
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (wide)
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<4>), dim3(grid), dim3(512), 0, s, (const __hip_bfloat16*)A,
                           (const __hip_bfloat16*)B, (__hip_bfloat16*)C, M, N, K, lda, ldb, ldc);
    else
        hipLaunchKernelGGL((gemm_nt_bf16_kernel<2>), dim3(grid), dim3(256), 0, s, (const __hip_bfloat16*)A,
                           (const __hip_bfloat16*)B, (__hip_bfloat16*)C, M, N, K, lda, ldb, ldc);
    return check_launch("gemm_nt_bf16_kernel");
}

// General entry point: see include/speech_hip.h.  The big K-contiguous bf16 problems go to the kernels
// above, everything else to the general kernel (gemm_generic.hip).
extern "C" int sp_gemm(const void* A, const void* B, void* C, int M, int N, int K, int64_t sam, int64_t sak,
                       int64_t sbn, int64_t sbk, int64_t ldc, int in_dtype, int out_dtype, int accumulate,
                       int split_k, int64_t c_split_stride, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "sp_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    SP_CHECK_ARG(in_dtype == SP_F32 || in_dtype == SP_F16 || in_dtype == SP_BF16, "sp_gemm: in_dtype %d", in_dtype);
    SP_CHECK_ARG(out_dtype == SP_F32 || out_dtype == SP_F16 || out_dtype == SP_BF16, "sp_gemm: out_dtype %d", out_dtype);
    SP_CHECK_ARG(sak == 1 || sam == 1, "sp_gemm: A must be contiguous along k (sak == 1) or along m (sam == 1)");
    SP_CHECK_ARG(sbk == 1 || sbn == 1, "sp_gemm: B must be contiguous along k (sbk == 1) or along n (sbn == 1)");
    SP_CHECK_ARG(!(sak == 1 && sam == 1 && M > 1 && K > 1) && !(sbk == 1 && sbn == 1 && N > 1 && K > 1),
                 "sp_gemm: overlapping operand strides");
    SP_CHECK_ARG(ldc >= N, "sp_gemm: ldc %lld < N %d", (long long)ldc, N);
    SP_CHECK_ARG(split_k >= 1 && split_k <= 64, "sp_gemm: split_k %d outside [1, 64]", split_k);
    SP_CHECK_ARG(!accumulate || out_dtype == SP_F32, "sp_gemm: accumulation needs an f32 output");
    SP_CHECK_ARG(split_k == 1 || c_split_stride >= (int64_t)M * ldc,
                 "sp_gemm: split-K slabs must not overlap (c_split_stride >= M * ldc)");
    if (M == 0 || N == 0) return SP_OK;
    SP_CHECK_ARG(C && (K == 0 || (A && B)), "sp_gemm: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int vec = in_dtype == SP_F32 ? 4 : 8;
    const int64_t a_ld = sak == 1 ? sam : sak, b_ld = sbk == 1 ? sbn : sbk;
    const bool aligned_in = (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && a_ld % vec == 0 && b_ld % vec == 0;
    // the fast kernels: both operands K-contiguous, bf16 in and out, whole tiles in N and K
    if (sak == 1 && sbk == 1 && in_dtype == SP_BF16 && out_dtype == SP_BF16 && !accumulate && split_k == 1 &&
        aligned_in && M >= 512 && N % 128 == 0 && K % kBK == 0 && K >= kBK && ldc % 4 == 0 && ((uintptr_t)C & 7) == 0 &&
        sam >= K && sbn >= K)
        return sp_gemm_nt_bf16(A, B, C, M, N, K, sam, sbn, ldc, stream);
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
    g.sam = sam; g.sak = sak; g.sbn = sbn; g.sbk = sbk; g.ldc = ldc;
    g.out_dtype = out_dtype; g.accumulate = accumulate; g.split_k = split_k; g.c_split_stride = c_split_stride;
    g.vec_in = aligned_in;
    SP_CHECK_ARG((int64_t)((M + 127) / 128) < 65536, "sp_gemm: M too large for the general kernel");
    return gemm_generic(g, in_dtype, s);
}
