// Fused magnitude-mel frontend for gfx950: framing (centre, reflect pad) + periodic Hann +
// 400-point real FFT + |.| + 201x80 mel projection + SpecAugment masks + zero padding, one pass.
// Replaces data.py:15-24 and the padding of data.py:174-175 of the reference (torchaudio
// MelSpectrogram on CPU DataLoader workers, one utterance at a time).
//
// Work decomposition: one WAVE owns 8 consecutive frames of one utterance and runs the transforms for them
// without a workgroup barrier; two waves (16 frames) then share the mel projection:
//   staging  the wave's 1520 samples, read once with coalesced 16-byte loads, into LDS
//   phase 1  200 lane-tasks (frame, n2): 16 windowed samples -> 8-point DFT -> W200 twiddle -> LDS (in place,
//            frame pairs from the last to the first)
//   phase 2   64 lane-tasks (frame, k1): 25-point DFT (5x5) entirely in registers -> LDS (in place)
//   phase 3  frame by frame, lane = bin k (and 64 + k): real-FFT split + magnitude for bins k and 200-k -> LDS
//   barrier  (the only one) -- both waves of a pair have their 8 x 201 magnitudes in LDS
//   phase 4  mel projection on the matrix cores: v_mfma_f32_16x16x4_f32 (exact f32), 16 mels x the pair's 16
//            frames per tile, only over the 4-bin blocks where the tile's filters are non-zero; the five tiles
//            are dealt to the pair's two waves by block count
//   phase 5  masked / padded store, [B][80][T_max], 64 bytes per mel row and tile
// Complex arithmetic is packed (one v_pk_*_f32 per butterfly, see below): at three waves per SIMD the kernel
// pays per instruction issued, not per flop.
// HBM traffic: every sample is read once from HBM (neighbouring waves' 240-sample overlap comes from L2),
// every output element written once.  LDS: 12.9 KB per wave (samples, intermediates and magnitudes alias),
// 3 blocks per CU.
#include <cstring>

#include "common.h"
#include "mel_math.h"

namespace sp {
using namespace melmath;

constexpr int kNfft = 400, kHop = 160, kBins = 201, kMels = 80;
constexpr int kF = 8;             // frames per wave
constexpr int kWPB = 4;           // waves per block
constexpr int kZStride = 200;     // complex points per frame (1600 B: the 25-point transforms' reads are conflict-free)
constexpr int kMagStride = 201;   // odd stride: conflict-free column reads
constexpr int kTiles = kMels / 16;
constexpr int kGroup = 4;         // 4-bin blocks per trip of the projection loop
constexpr int kMaxBlk = kTiles * 52 + 6 * kGroup;
constexpr int kStage = (kF - 1) * kHop + kNfft;   // samples a wave's 8 frames touch: 1520

struct MelPlan {
    float win[kNfft];
    cf tw200[25][8];   // W200^(n2*k1)
    cf tw400[104];     // W400^k, k = 0..100
    // The five 16-mel tiles are dealt to the two waves of a pair (each wave projects its tiles for the pair's 16
    // frames): slots [0, split) belong to the even wave, [split, kTiles) to the odd one, balanced by block count.
    int slot_tile[8];  // tile id of the slot
    int slot_nblk[8];  // 4-bin blocks of the tile, a multiple of kGroup (zero rows pad it)
    int slot_k0[8];    // first bin of the tile's first block
    int slot_blk0[8];  // first image of the slot in fbt
    int split;
    int share_groups[2];   // groups of kGroup blocks in each wave's share, even
    int pad_;
    // A operand images in slot order, two groups of all-zero blocks after each wave's share (the loop requests
    // two groups ahead): lane l -> fb[k0 + 4 blk + (l>>4)][16 tile + (l&15)]
    float fbt[kMaxBlk][64];
};

using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
// Every LDS access of the kernel goes through these types: the three views of a wave's buffer (staged samples,
// complex intermediates, magnitudes) overlap on purpose, so the compiler must keep their program order.
typedef float __attribute__((may_alias)) lds_f1;
typedef f32x2 __attribute__((may_alias)) lds_f2;
typedef f32x4 __attribute__((may_alias)) lds_f4;
typedef f32x4 __attribute__((may_alias, aligned(4))) gmem_f4;   // 16-byte load of 4-byte aligned samples

// Per-wave LDS, 12 800 (+ 64 of padding) bytes, three views one after the other in time:
//   smp  the 1520 samples the 8 frames touch (bytes 0..6080), read by phase 1 frame pair by frame pair from the
//        LAST pair to the first, so that the pair's 8-point transforms can go to
//   yz   [frame][200] complex (1600 bytes per frame) without touching samples a later pass still needs;
//        phase 2 transforms it in place;
//   mag  [frame][201] magnitudes (804 bytes per frame): frame f's magnitudes land on yz of frames <= f / 2,
//        which phase 3 has consumed by then (frame 0: all four reads are issued before the first write, and
//        one wave's LDS operations execute in order).
constexpr int kWaveLdsBytes = kF * kZStride * 8 + 64;   // + 16 banks: the pair's two magnitude images do not collide
static_assert(2 * kZStride * 8 >= (kHop + kNfft) * 4, "pass 1 writes above the samples pass 0 still needs");
static_assert(kWaveLdsBytes % 16 == 0, "16-byte staging stores");

// ---- packed complex arithmetic ------------------------------------------------------------------------
// A complex number is a register pair (re, im) and every butterfly is ONE v_pk_*_f32: a wave issues an
// instruction every ~5 cycles whatever it is (tools/ubench/valu_tput.hip), so at the three waves per SIMD the
// LDS footprint allows, the instruction count, not the flop count, is what the transform costs.  The compiler
// packs plain adds and scalar multiples by itself; multiplications by +-i and conjugations need the
// op_sel / neg modifiers, which it does not fold (it emits v_xor + v_mov), hence the asm helpers.  The scalar
// forms of the same transforms live in mel_math.h (host replay: tests/host/mel_emulate.cpp).
using c2 = f32x2;
#define SP_PK2(name, text)                                        \
    __device__ __forceinline__ c2 name(c2 a, c2 b) {              \
        c2 d;                                                     \
        asm(text : "=v"(d) : "v"(a), "v"(b));                     \
        return d;                                                 \
    }
SP_PK2(add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")     // a + (-i) b
SP_PK2(add_pi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")     // a + i b
SP_PK2(add_conj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")                                // a + conj(b)
SP_PK2(sub_conj, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]")                                // a - conj(b)
#undef SP_PK2
// a * w: (a.im w.im, a.im w.re), then (a.re w.re - ., a.re w.im + .)
__device__ __forceinline__ c2 cmul(c2 a, c2 w) {
    c2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(d) : "v"(a), "v"(w), "v"(t));
    return d;
}
// the same with a compile-time constant w held in a scalar register pair
__device__ __forceinline__ c2 cmul_k(c2 a, float wre, float wim) {
    const c2 w = {wre, wim};
    c2 t, d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "s"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1] neg_lo:[0,0,1]" : "=v"(d) : "v"(a), "s"(w), "v"(t));
    return d;
}

// forward 8-point DFT in place (mel_math.h dft8)
__device__ __forceinline__ void dft8_pk(c2 (&z)[8]) {
    const c2 a0 = z[0] + z[4], a4 = z[0] - z[4], a1 = z[1] + z[5], a5 = z[1] - z[5];
    const c2 a2 = z[2] + z[6], a6 = z[2] - z[6], a3 = z[3] + z[7], a7 = z[3] - z[7];
    const c2 b0 = a0 + a2, b2 = a0 - a2, b1 = a1 + a3, b3 = a1 - a3;
    z[0] = b0 + b1; z[4] = b0 - b1; z[2] = add_mi(b2, b3); z[6] = add_pi(b2, b3);
    const c2 c1 = add_mi(a5, a5) * kInvSqrt2;      // a5 (1 - i) / sqrt 2
    const c2 c3 = add_pi(a7, a7) * -kInvSqrt2;     // a7 (-1 - i) / sqrt 2
    const c2 d0 = add_mi(a4, a6), d2 = add_pi(a4, a6);
    const c2 d1 = c1 + c3, d3 = c1 - c3;
    z[1] = d0 + d1; z[5] = d0 - d1; z[3] = add_mi(d2, d3); z[7] = add_pi(d2, d3);
}

// forward 5-point DFT (mel_math.h dft5): 18 packed instructions
__device__ __forceinline__ void dft5_pk(c2 x0, c2 x1, c2 x2, c2 x3, c2 x4, c2& y0, c2& y1, c2& y2, c2& y3, c2& y4) {
    const c2 t1 = x1 + x4, t2 = x2 + x3, t3 = x1 - x4, t4 = x2 - x3;
    y0 = x0 + t1 + t2;
    const c2 a1 = x0 + kC1 * t1 + kC2 * t2, a2 = x0 + kC2 * t1 + kC1 * t2;
    const c2 b1 = kS1 * t3 + kS2 * t4, b2 = kS2 * t3 - kS1 * t4;
    y1 = add_mi(a1, b1); y4 = add_pi(a1, b1);
    y2 = add_mi(a2, b2); y3 = add_pi(a2, b2);
}

// forward 25-point DFT in place (mel_math.h dft25)
__device__ __forceinline__ void dft25_pk(c2 (&y)[25]) {
    constexpr float twre[4][4] = SP_TW25_RE;
    constexpr float twim[4][4] = SP_TW25_IM;
    c2 g[5][5];
#pragma unroll
    for (int b = 0; b < 5; ++b) dft5_pk(y[b], y[5 + b], y[10 + b], y[15 + b], y[20 + b], g[b][0], g[b][1], g[b][2], g[b][3], g[b][4]);
#pragma unroll
    for (int b = 1; b < 5; ++b)
#pragma unroll
        for (int c = 1; c < 5; ++c) g[b][c] = cmul_k(g[b][c], twre[b - 1][c - 1], twim[b - 1][c - 1]);
#pragma unroll
    for (int c = 0; c < 5; ++c) dft5_pk(g[0][c], g[1][c], g[2][c], g[3][c], g[4][c], y[c], y[c + 5], y[c + 10], y[c + 15], y[c + 20]);
}

// real-FFT split for one bin pair (mel_math.h real_post) WITHOUT its two halvings: from Zk = Z[k], Zm = Z[200-k]
// and w = W400^k, 2|X[k]| and 2|X[200-k]| (the filterbank images carry the factor 1/2, exactly)
__device__ __forceinline__ void real_post_pk(c2 Zk, c2 Zm, c2 w, float& mag_k, float& mag_m) {
    const c2 E = add_conj(Zk, Zm), P = cmul(sub_conj(Zk, Zm), w);
    const c2 xa = add_mi(E, P), xb = add_pi(E, P);
    mag_k = __builtin_amdgcn_sqrtf(fmaf(xa.y, xa.y, xa.x * xa.x));
    mag_m = __builtin_amdgcn_sqrtf(fmaf(xb.y, xb.y, xb.x * xb.x));
}

__device__ __forceinline__ c2 ld2(const char* p) { return *reinterpret_cast<const lds_f2*>(p); }
__device__ __forceinline__ void st2(char* p, c2 v) { *reinterpret_cast<lds_f2*>(p) = v; }
__device__ __forceinline__ void st1(char* p, float v) { *reinterpret_cast<lds_f1*>(p) = v; }
__device__ __forceinline__ c2 ldg2(const void* p) { return *reinterpret_cast<const c2*>(p); }

// S = sample type: float in [-1, 1], or int16_t PCM read as sample / 32768 (exact in f32: what
// torchaudio.load(normalize=True) hands the reference for 16-bit files) -- half the bytes to read.
__device__ __forceinline__ float pcm(const float* x, int i) { return x[i]; }
__device__ __forceinline__ float pcm(const int16_t* x, int i) { return (float)x[i] * (1.f / 32768.f); }

// Staging and phases 1-3 for one wave: samples of 8 frames -> their 201 magnitudes each, in the wave's LDS.
template <typename S>
__device__ __forceinline__ void transform_frames(const S* __restrict__ x, int N, int t0, const MelPlan* __restrict__ plan,
                                                 char* L, int lane) {
    // ------------------------------------------------------------------ staging
    // The wave's 1520 samples, [t0*160 - 200, (t0+7)*160 + 200), are read ONCE, coalesced, into LDS; the
    // 2.5-fold frame overlap and the stride-50 gathers of the 8-point transforms are served from there.
    // Waves at an utterance's ends (reflection, frames past the end) take the element-wise path.
    // (int16 PCM: the same, 8 bytes per lane and round.)  The lane's window taps and twiddles of phases 1 and 3 are requested first (L2 hits): the wait for the
    // samples covers them.
    const int fh = lane >= 25 ? 1 : 0;
    const int n2 = min(lane - 25 * fh, 24);
    const int kA = lane, kB = min(64 + lane, 100);
    c2 w[8], tw[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) w[n1] = ldg2(&plan->win[50 * n1 + 2 * n2]);
#pragma unroll
    for (int k1 = 1; k1 < 8; ++k1) tw[k1] = ldg2(&plan->tw200[n2][k1]);
    const c2 twA = ldg2(&plan->tw400[kA]), twB = ldg2(&plan->tw400[kB]);

    const int s0 = t0 * kHop - kNfft / 2;
    const bool interior = s0 >= 0 && s0 + kStage <= N;
    if (sizeof(S) == 4 && interior) {
        const gmem_f4* src = reinterpret_cast<const gmem_f4*>(x + s0);
        f32x4 v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = src[min(lane + 64 * i, kStage / 4 - 1)];
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (lane + 64 * i < kStage / 4) reinterpret_cast<lds_f4*>(L)[lane + 64 * i] = v[i];
    } else if (sizeof(S) == 2 && s0 >= 0 && s0 + kStage + 2 <= N) {
        // int16 PCM, interior wave: 4 samples per lane and round from whole dwords.  The first sample may sit in
        // the upper half of its dword (utterance offsets are in samples): sh = 0 / 1 halfwords, three dwords cover
        // the lane's four samples either way and v_alignbit shifts them into place.
        const uintptr_t a0 = reinterpret_cast<uintptr_t>(x + s0);
        const int sh = (int)(a0 >> 1) & 1;
        typedef uint32_t __attribute__((aligned(4))) u32a;
        typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
        typedef u32x2 __attribute__((aligned(4))) u32x2a;
        const u32a* q = reinterpret_cast<const u32a*>(a0 - 2 * sh);
        u32x2 d01[6];
        uint32_t d2[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g = min(lane + 64 * i, kStage / 4 - 1);
            d01[i] = *reinterpret_cast<const u32x2a*>(q + 2 * g);
            d2[i] = q[2 * g + 2];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const uint32_t w0 = __builtin_amdgcn_alignbit(d01[i].y, d01[i].x, 16 * sh);
            const uint32_t w1 = __builtin_amdgcn_alignbit(d2[i], d01[i].y, 16 * sh);
            f32x4 v;
            v.x = (float)(int16_t)(w0 & 0xffff) * (1.f / 32768.f);
            v.y = (float)((int32_t)w0 >> 16) * (1.f / 32768.f);
            v.z = (float)(int16_t)(w1 & 0xffff) * (1.f / 32768.f);
            v.w = (float)((int32_t)w1 >> 16) * (1.f / 32768.f);
            if (lane + 64 * i < kStage / 4) reinterpret_cast<lds_f4*>(L)[lane + 64 * i] = v;
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < (kStage + 63) / 64; ++i) {
            const int j = lane + 64 * i;
            int idx = interior ? s0 + j : reflect(s0 + j, N);
            idx = min(max(idx, 0), N - 1);
            const float v = pcm(x, idx);
            if (j < kStage) st1(L + 4 * j, v);
        }
    }

    // ------------------------------------------------------------------ phase 1
    // 200 lane-tasks (frame, n2): lane = (frame parity, n2) and 4 passes walk the frame pairs from the last to
    // the first (see the LDS note above); a lane's 16 window taps and 7 twiddles are loaded once.
    {
        const bool active = lane < 50;
        const char* src = L + fh * (kHop * 4) + n2 * 8;
        char* dst = L + fh * (kZStride * 8) + n2 * 8;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) {
            const int p = 3 - pp;
            c2 z[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) z[n1] = ld2(src + p * (2 * kHop * 4) + n1 * 200) * w[n1];
            dft8_pk(z);
            if (active) {
                st2(dst + p * (2 * kZStride * 8), z[0]);
#pragma unroll
                for (int k1 = 1; k1 < 8; ++k1) st2(dst + p * (2 * kZStride * 8) + k1 * 200, cmul(z[k1], tw[k1]));
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ------------------------------------------------------------------ phase 2
    {
        const int f = lane >> 3, k1 = lane & 7;
        char* base = L + f * (kZStride * 8);
        c2 y[25];
#pragma unroll
        for (int n2 = 0; n2 < 25; ++n2) y[n2] = ld2(base + (k1 * 25 + n2) * 8);
        dft25_pk(y);
#pragma unroll
        for (int k2 = 0; k2 < 25; ++k2) st2(base + (k1 + 8 * k2) * 8, y[k2]);
    }
    __builtin_amdgcn_wave_barrier();

    // ------------------------------------------------------------------ phase 3
    // Frame by frame, lane = bin: k = lane and k = 64 + lane (37 lanes) with their mirror bins 200 - k; the
    // next frame's four points are read before this frame's magnitudes are written.
    {
        const bool actB = lane < 37;
        const int mA = (kA == 0 ? 0 : 200 - kA) * 8;   // Z[200] := Z[0]
        c2 zkA = ld2(L + kA * 8), zmA = ld2(L + mA);
        c2 zkB = ld2(L + kB * 8), zmB = ld2(L + (200 - kB) * 8);
#pragma unroll
        for (int f = 0; f < kF; ++f) {
            const c2 ckA = zkA, cmA = zmA, ckB = zkB, cmB = zmB;
            if (f + 1 < kF) {
                const char* nb = L + (f + 1) * (kZStride * 8);
                zkA = ld2(nb + kA * 8); zmA = ld2(nb + mA);
                zkB = ld2(nb + kB * 8); zmB = ld2(nb + (200 - kB) * 8);
            }
            float m0, m1, m2, m3;
            real_post_pk(ckA, cmA, twA, m0, m1);
            real_post_pk(ckB, cmB, twB, m2, m3);
            char* mrow = L + f * (kMagStride * 4);
            st1(mrow + kA * 4, m0);
            st1(mrow + (200 - kA) * 4, m1);
            if (actB) {
                st1(mrow + kB * 4, m2);
                st1(mrow + (200 - kB) * 4, m3);
            }
        }
    }
}

template <typename S>
__global__ __launch_bounds__(64 * kWPB, 3) void melspec_kernel(
    const S* __restrict__ wav, const int64_t* __restrict__ offs, int T_max,
    const MelPlan* __restrict__ plan, const int32_t* __restrict__ masks, int flags,
    float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) char s_lds[kWPB][kWaveLdsBytes];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: frame ranges, slots and the plan's
    char* L = s_lds[wave];                                               // fields stay in scalar registers
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * kWPB + wave) * kF;
    const int tp = (blockIdx.x * kWPB + (wave & ~1)) * kF;   // first frame of the wave pair
    // A pair leaves together (a finished wave no longer counts at the workgroup barrier below).
    if (tp >= T_max) return;
    const int64_t off = offs[b];
    const int N = (int)(offs[b + 1] - off);
    const int Tb = N > 0 ? 1 + N / kHop : 0;
    const S* x = wav + off;
    float* outb = out + (int64_t)b * kMels * T_max;

    if (tp >= Tb) {  // pure padding: zeros
        for (int i = lane; i < kMels * kF; i += 64) {
            const int m = i >> 3, f = i & 7;
            if (t0 + f < T_max) outb[(int64_t)m * T_max + t0 + f] = 0.f;
        }
        return;
    }
    // A wave past the utterance's (or the batch's) last frame beside a live partner skips the transforms: its
    // columns of the projection hold whatever its LDS rows held and are masked / not stored.
    const bool live = t0 < Tb && t0 < T_max;

    int mf0 = 0, mfw = 0, mt0 = 0, mtw = 0;
    if (masks) {
        mf0 = masks[b * 4 + 0]; mfw = masks[b * 4 + 1];
        mt0 = masks[b * 4 + 2]; mtw = masks[b * 4 + 3];
    }
    // this wave's share of the projection: plan->share_groups[par] groups from slot slot_beg on; its first group of filterbank images is
    // requested here, three phases ahead of its use
    const int par = wave & 1;
    const int slot_beg = par ? plan->split : 0;
    const float* a_ptr = plan->fbt[plan->slot_blk0[slot_beg]] + lane;
    float a_p[kGroup], a_q[kGroup];   // images of the group being multiplied and of the one after it
#pragma unroll
    for (int r = 0; r < kGroup; ++r) a_p[r] = a_ptr[64 * r];
#pragma unroll
    for (int r = 0; r < kGroup; ++r) a_q[r] = a_ptr[64 * (kGroup + r)];

    if (live) transform_frames(x, N, t0, plan, L, lane);
    __syncthreads();   // both magnitude images of the pair are complete

    // ------------------------------------------------------------------ phases 4 + 5
    // Mel projection on the matrix cores: a tile is 16 mels x the PAIR's 16 frames (columns 0-7 from the even
    // wave's magnitudes, 8-15 from the odd wave's), over the tile's non-zero 4-bin blocks only: per block one
    // coalesced 256-byte image load (requested two groups ahead; a wave's images are contiguous, so the request runs
    // across tile boundaries), one LDS read and one v_mfma_f32_16x16x4_f32; then the tile's 16 x 16 outputs are
    // masked and stored, 64 bytes per mel row.
    {
        const int fcol = lane & 15, kq = lane >> 4;
        const char* magrow = s_lds[(wave & ~1) + (fcol >> 3)] + (fcol & 7) * (kMagStride * 4) + kq * 4;
        const int t = tp + fcol;
        const bool tmask = (t >= Tb) || (t >= mt0 && t < mt0 + mtw);
        const bool store_lane = t < T_max;
        // one unsigned compare per value: (mel - mf0) < mfw; a masked frame makes every mel fall inside
        const int melb = kq * 4 - (tmask ? -(1 << 20) : mf0);
        const unsigned mlim = tmask ? 0x7fffffffu : (unsigned)mfw;
        float* orow = outb + (int64_t)(kq * 4) * T_max + t;
        // The wave's share is one flat run of groups (an even number of them: the plan pads); a group's images
        // are replaced, as soon as its four MFMAs are issued, by those of the group two further on, so two
        // register sets alternate without a copy (a copy would wait for the load it moves).
        int slot = slot_beg;
        const int ngroups = plan->share_groups[par];
        int left = plan->slot_nblk[slot] / kGroup;
        const char* bp = magrow + plan->slot_k0[slot] * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        auto group = [&](float (&a)[kGroup]) {
            float bv[kGroup];
#pragma unroll
            for (int r = 0; r < kGroup; ++r) bv[r] = *reinterpret_cast<const lds_f1*>(bp + 16 * r);
            bp += 16 * kGroup;
#pragma unroll
            for (int r = 0; r < kGroup; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], bv[r], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < kGroup; ++r) a[r] = a_ptr[64 * (2 * kGroup + r)];
            a_ptr += 64 * kGroup;
            if (--left == 0) {   // the tile is complete: mask, store, next slot
                const int tile = plan->slot_tile[slot];
                if (store_lane) {
                    float* o = orow + (int64_t)(tile * 16) * T_max;
                    if (flags & 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = logf(acc[r] + 1e-6f);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[(int64_t)r * T_max] = (unsigned)(melb + tile * 16 + r) < mlim ? 0.f : acc[r];
                }
                acc = f32x4{0.f, 0.f, 0.f, 0.f};
                ++slot;   // past the share's last slot the plan holds a zero-length entry that is never reached
                left = plan->slot_nblk[slot] / kGroup;
                bp = magrow + plan->slot_k0[slot] * 4;
            }
        };
        for (int g = 0; g < ngroups; g += 2) {
            group(a_p);
            group(a_q);
        }
    }
}

// Per-utterance normalisation (flags bit 1), a light second pass over the features the first pass left in
// L2 / HBM: one wave per (utterance, mel row), mean and biased variance over the utterance's T_b valid frames
// taken in two sweeps of the row (sum, then sum of squared deviations: no cancellation), y = (x - mean) *
// rsqrt(var + 1e-5); SpecAugment bands are applied here (the statistics see the unmasked features), padding
// frames stay 0.
__global__ __launch_bounds__(256) void melnorm_kernel(float* __restrict__ out, const int64_t* __restrict__ offs, int T_max,
                                                        const int32_t* __restrict__ masks) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int mel = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (mel >= kMels) return;
    const int Tb = min((int)((offs[b + 1] - offs[b]) / kHop) + 1, T_max);
    float* row = out + ((int64_t)b * kMels + mel) * T_max;
    float s = 0.f;
    for (int t = lane; t < Tb; t += 64) s += row[t];
    const float mean = wave_sum(s) / (float)Tb;
    float q = 0.f;
    for (int t = lane; t < Tb; t += 64) {
        const float dlt = row[t] - mean;
        q += dlt * dlt;
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)Tb + 1e-5f);
    int mf0 = 0, mfw = 0, mt0 = 0, mtw = 0;
    if (masks) { mf0 = masks[4 * b]; mfw = masks[4 * b + 1]; mt0 = masks[4 * b + 2]; mtw = masks[4 * b + 3]; }
    const bool frow = mel >= mf0 && mel < mf0 + mfw;
    for (int t = lane; t < Tb; t += 64) {
        const bool m = frow || (t >= mt0 && t < mt0 + mtw);
        row[t] = m ? 0.f : (row[t] - mean) * rstd;
    }
}

}  // namespace sp

extern "C" size_t sp_melspec_plan_bytes(void) { return sizeof(sp::MelPlan); }

extern "C" int sp_melspec_plan(const float* fbank_host, void* plan_dev, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(fbank_host && plan_dev, "sp_melspec_plan: null pointer");
    static thread_local MelPlan hp;  // ~70 KB staging copy; hipMemcpyAsync from pageable memory
                                     // completes the host read before returning
    std::memset(&hp, 0, sizeof(hp));
    const double two_pi = 6.283185307179586476925286766559;
    for (int n = 0; n < kNfft; ++n) hp.win[n] = (float)(0.5 - 0.5 * std::cos(two_pi * n / kNfft));
    for (int n2 = 0; n2 < 25; ++n2)
        for (int k1 = 0; k1 < 8; ++k1) {
            const double a = two_pi * (double)(n2 * k1) / 200.0;
            hp.tw200[n2][k1] = melmath::mk((float)std::cos(a), (float)-std::sin(a));
        }
    for (int k = 0; k <= 100; ++k) {
        const double a = two_pi * (double)k / 400.0;
        hp.tw400[k] = melmath::mk((float)std::cos(a), (float)-std::sin(a));
    }
    // per tile: the bins its 16 filters touch, in whole groups of 4-bin blocks; the window is moved down rather
    // than past the last bin where it can be (bins 201..203 of a row are finite leftovers that meet zero rows)
    int t_k0[kTiles], t_nblk[kTiles];
    for (int tile = 0; tile < kTiles; ++tile) {
        int lo = kBins, hi = -1;
        for (int f = 0; f < kBins; ++f)
            for (int m = 16 * tile; m < 16 * tile + 16; ++m)
                if (fbank_host[f * kMels + m] != 0.f) {
                    lo = f < lo ? f : lo;
                    hi = f > hi ? f : hi;
                }
        int nblk = hi < 0 ? 1 : (hi - lo + 4) / 4;   // a tile without any weight still stores its zeros
        nblk = (nblk + kGroup - 1) / kGroup * kGroup;
        int k0 = hi < 0 ? 0 : lo;
        if (k0 + 4 * nblk > kBins) k0 = kBins - 4 * nblk > 0 ? kBins - 4 * nblk : 0;
        t_k0[tile] = k0;
        t_nblk[tile] = nblk;
    }
    // deal the tiles to the pair's two waves: the subset whose block count is nearest to half
    int best = 0, best_diff = 1 << 30, total = 0;
    for (int tile = 0; tile < kTiles; ++tile) total += t_nblk[tile];
    for (int sub = 0; sub < (1 << kTiles); ++sub) {
        int n = 0;
        for (int tile = 0; tile < kTiles; ++tile) n += (sub >> tile & 1) ? t_nblk[tile] : 0;
        const int diff = n * 2 > total ? n * 2 - total : total - n * 2;
        if (diff < best_diff) { best_diff = diff; best = sub; }
    }
    int blk = 0, slot = 0;
    for (int side = 0; side < 2; ++side) {
        const int blk_side = blk, slot_side = slot;
        for (int tile = 0; tile < kTiles; ++tile) {
            if (((best >> tile) & 1) != (side == 0)) continue;
            hp.slot_tile[slot] = tile;
            hp.slot_k0[slot] = t_k0[tile];
            hp.slot_nblk[slot] = t_nblk[tile];
            hp.slot_blk0[slot] = blk;
            for (int kb = 0; kb < t_nblk[tile]; ++kb, ++blk)
                for (int l = 0; l < 64; ++l) {
                    const int f = t_k0[tile] + 4 * kb + (l >> 4), m = 16 * tile + (l & 15);
                    hp.fbt[blk][l] = f < kBins ? 0.5f * fbank_host[f * kMels + m] : 0.f;   // phase 3 leaves 2|X|
                }
            ++slot;
        }
        // an even number of groups per share: one more group of zero images for the share's last tile
        if (slot > slot_side && ((blk - blk_side) / kGroup) % 2) {
            hp.slot_nblk[slot - 1] += kGroup;
            blk += kGroup;
        }
        hp.share_groups[side] = (blk - blk_side) / kGroup;
        if (side == 0) hp.split = slot;
        blk += 2 * kGroup;   // zero images the look-ahead of the share's last groups lands on
    }
    for (; slot < 8; ++slot) hp.slot_blk0[slot] = blk - 2 * kGroup;   // an empty share still requests two (zero) groups
    hipError_t e = hipMemcpyAsync(plan_dev, &hp, sizeof(hp), hipMemcpyHostToDevice,
                                  reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(SP_ELAUNCH, "sp_melspec_plan: %s", hipGetErrorString(e));
    e = hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream));  // hp is reused by the next call
    if (e != hipSuccess) return fail(SP_ELAUNCH, "sp_melspec_plan: %s", hipGetErrorString(e));
    return SP_OK;
}

extern "C" int sp_melspec_fwd(const void* wav, const int64_t* wav_offsets, int B, int T_max,
                              const void* plan_dev, const int32_t* masks, int flags, float* out,
                              void* stream) {
    using namespace sp;
    SP_CHECK_ARG(B >= 0 && T_max >= 0, "sp_melspec_fwd: bad shape B=%d T_max=%d", B, T_max);
    if (B == 0 || T_max == 0) return SP_OK;
    SP_CHECK_ARG(wav && wav_offsets && plan_dev && out, "sp_melspec_fwd: null pointer");
    SP_CHECK_ARG(B <= 65535, "sp_melspec_fwd: B=%d exceeds grid limit", B);
    SP_CHECK_ARG((flags & ~7) == 0, "sp_melspec_fwd: unknown flag bits in %d", flags);
    const int per_block = kF * kWPB;
    dim3 grid((T_max + per_block - 1) / per_block, B);
    // with the normalisation pass the SpecAugment bands are applied there, after the statistics
    const int32_t* masks1 = (flags & 2) ? nullptr : masks;
    if (flags & 4)
        hipLaunchKernelGGL((melspec_kernel<int16_t>), grid, dim3(64 * kWPB), 0, reinterpret_cast<hipStream_t>(stream),
                           (const int16_t*)wav, wav_offsets, T_max, reinterpret_cast<const MelPlan*>(plan_dev), masks1,
                           flags, out);
    else
        hipLaunchKernelGGL((melspec_kernel<float>), grid, dim3(64 * kWPB), 0, reinterpret_cast<hipStream_t>(stream),
                           (const float*)wav, wav_offsets, T_max, reinterpret_cast<const MelPlan*>(plan_dev), masks1,
                           flags, out);
    int rc = check_launch("melspec_kernel");
    if (rc || !(flags & 2)) return rc;
    hipLaunchKernelGGL(melnorm_kernel, dim3((kMels + 3) / 4, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), out,
                       wav_offsets, T_max, masks);
    return check_launch("melnorm_kernel");
}
