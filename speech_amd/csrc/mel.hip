// Fused magnitude-mel frontend for gfx950: framing (centre, reflect pad) + periodic Hann +
// 400-point real FFT + |.| + 201x80 mel projection + SpecAugment masks + zero padding, one pass.
// Replaces data.py:15-24 and the padding of data.py:174-175 of the reference (torchaudio
// MelSpectrogram on CPU DataLoader workers, one utterance at a time).
//
// Work decomposition: one WAVE owns 8 consecutive frames of one utterance and runs the whole
// chain for them without any workgroup barrier (4 independent waves per 256-thread block, so
// waves drift apart and overlap each other's load / VALU / LDS phases):
//   phase 1  200 lane-tasks (frame, n2): 16 windowed samples -> 8-point DFT -> W200 twiddle -> LDS
//   phase 2   64 lane-tasks (frame, k1): 25-point DFT (5x5) entirely in registers -> LDS (in place)
//   phase 3  808 lane-tasks (frame, k<=100): real-FFT split + magnitude for bins k and 200-k -> LDS
//   phase 4  mel projection on the matrix cores: v_mfma_f32_16x16x4_f32 (exact f32), mel x frame
//            tiles, only over the 4-bin blocks where the 16-mel tile's filters are non-zero
//   phase 5  masked / padded store, [B][80][T_max]
// HBM traffic: every sample is read once from HBM (the 2.5x frame overlap is served by L1/L2),
// every output element written once.  LDS: 12.8 KB per wave (the magnitudes alias the FFT buffer), 3 blocks per CU.
#include <cstring>

#include "common.h"
#include "mel_math.h"

namespace sp {
using namespace melmath;

constexpr int kNfft = 400, kHop = 160, kBins = 201, kMels = 80;
constexpr int kF = 8;           // frames per wave
constexpr int kWPB = 4;         // waves per block
constexpr int kMagStride = 201; // odd stride: conflict-free column reads
constexpr int kTiles = kMels / 16;
constexpr int kMaxBlk = kTiles * 51;

struct MelPlan {
    float win[kNfft];
    cf tw200[25][8];   // W200^(n2*k1)
    cf tw400[104];     // W400^k, k = 0..100
    int tile_blk0[8];  // first block of tile in fbt
    int tile_nblk[8];
    int tile_k0[8];    // first bin of tile
    float fbt[kMaxBlk][64];  // A operand images: lane l -> fb[k0 + 4 blk + (l>>4)][16 tile + (l&15)]
};

// Per-wave LDS: the 200-point complex intermediate of 8 frames; after phase 3 has pulled it into
// registers the same bytes hold the 8 x 201 magnitudes (12.8 KB per wave, 51 KB per block, three
// blocks per CU).
struct WaveLds {
    union {
        cf yz[kF][200];
        float mag[kF][kMagStride];
    };
};

using f32x4 = __attribute__((ext_vector_type(4))) float;

// The kernel is latency-bound, not issue-bound (one wave runs ~2.7 k instructions on 8 frames, the
// chip holds 3 waves per SIMD): every phase therefore issues all of its loads before its first use
// (straight-line code, registers instead of occupancy), and the mel projection runs its A-operand
// loads eight MFMAs ahead in a register ring.
// S = sample type: float in [-1, 1], or int16_t PCM read as sample / 32768 (exact in f32: what
// torchaudio.load(normalize=True) hands the reference for 16-bit files) -- half the bytes to read.
__device__ __forceinline__ float pcm(const float* x, int i) { return x[i]; }
__device__ __forceinline__ float pcm(const int16_t* x, int i) { return (float)x[i] * (1.f / 32768.f); }

template <typename S>
__global__ __launch_bounds__(64 * kWPB, 3) void melspec_kernel(
    const S* __restrict__ wav, const int64_t* __restrict__ offs, int T_max,
    const MelPlan* __restrict__ plan, const int32_t* __restrict__ masks, int flags,
    float* __restrict__ out) {
    __shared__ WaveLds s_lds[kWPB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveLds& L = s_lds[wave];
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * kWPB + wave) * kF;
    if (t0 >= T_max) return;
    const int64_t off = offs[b];
    const int N = (int)(offs[b + 1] - off);
    const int Tb = N > 0 ? 1 + N / kHop : 0;
    const S* x = wav + off;
    float* outb = out + (int64_t)b * kMels * T_max;

    if (t0 >= Tb) {  // pure padding: zeros
        for (int i = lane; i < kMels * kF; i += 64) {
            const int m = i >> 3, f = i & 7;
            if (t0 + f < T_max) outb[(int64_t)m * T_max + t0 + f] = 0.f;
        }
        return;
    }

    // ------------------------------------------------------------------ phase 1
    // 200 lane-tasks (frame, n2): lane = (frame parity, n2) and 4 passes walk the frame pairs, so a
    // lane's 16 window taps and 7 twiddles are loaded once; the 16 samples of ALL passes are
    // requested before the first butterfly (interior waves: plain indices, no reflection).
    int mf0 = 0, mfw = 0, mt0 = 0, mtw = 0;
    if (masks) {
        mf0 = masks[b * 4 + 0]; mfw = masks[b * 4 + 1];
        mt0 = masks[b * 4 + 2]; mtw = masks[b * 4 + 3];
    }
    const bool interior = (t0 * kHop - kNfft / 2 >= 0) && ((t0 + kF - 1) * kHop + kNfft / 2 <= N) &&
                          (t0 + kF <= Tb);
    {
        const int fh = lane >= 25 ? 1 : 0;
        const int n2 = min(lane - 25 * fh, 24);
        const bool active = lane < 50;
        float xs[4][16];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int t = t0 + 2 * p + fh;
            const int base = t * kHop - kNfft / 2 + 2 * n2;
            if (interior) {
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    xs[p][2 * n1] = pcm(x, base + 50 * n1);
                    xs[p][2 * n1 + 1] = pcm(x, base + 50 * n1 + 1);
                }
            } else {
                const bool live = t < Tb;
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    int i0 = reflect(base + 50 * n1, N), i1 = reflect(base + 50 * n1 + 1, N);
                    i0 = min(max(i0, 0), N - 1);
                    i1 = min(max(i1, 0), N - 1);
                    const float v0 = pcm(x, i0), v1 = pcm(x, i1);
                    xs[p][2 * n1] = live ? v0 : 0.f;
                    xs[p][2 * n1 + 1] = live ? v1 : 0.f;
                }
            }
        }
        float2 w[8];
        cf tw[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) w[n1] = *reinterpret_cast<const float2*>(&plan->win[50 * n1 + 2 * n2]);
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) tw[k1] = plan->tw200[n2][k1];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int f = 2 * p + fh;
            cf z[8];
#pragma unroll
            for (int n1 = 0; n1 < 8; ++n1) z[n1] = mk(xs[p][2 * n1] * w[n1].x, xs[p][2 * n1 + 1] * w[n1].y);
            dft8(z);
            if (active) {
                L.yz[f][n2] = z[0];
#pragma unroll
                for (int k1 = 1; k1 < 8; ++k1) L.yz[f][k1 * 25 + n2] = mul(z[k1], tw[k1]);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ------------------------------------------------------------------ phase 2
    {
        const int f = lane >> 3, k1 = lane & 7;
        cf y[25];
#pragma unroll
        for (int n2 = 0; n2 < 25; ++n2) y[n2] = L.yz[f][k1 * 25 + n2];
        dft25(y);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k2 = 0; k2 < 25; ++k2) L.yz[f][k1 + 8 * k2] = y[k2];
    }
    __builtin_amdgcn_wave_barrier();

    // ------------------------------------------------------------------ phase 3
    // 808 lane-tasks (frame, k <= 100) in 13 passes.  Every pass's two spectrum points and its
    // twiddle are read first; only then do the magnitudes overwrite the same LDS bytes.
    {
        constexpr int NP = 13;
        cf zk[NP], zm[NP], tw[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int task = min(q * 64 + lane, kF * 101 - 1);
            const int f = task / 101, k = task - f * 101;
            const int km = (k == 0) ? 0 : 200 - k;
            zk[q] = L.yz[f][k];
            zm[q] = L.yz[f][km];
            tw[q] = plan->tw400[k];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all reads landed before the aliasing writes
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int task = q * 64 + lane;
            const int tk = min(task, kF * 101 - 1);
            const int f = tk / 101, k = tk - f * 101;
            float mk_, mm_;
            real_post(zk[q], zm[q], tw[q], mk_, mm_);
            if (task < kF * 101) {
                L.mag[f][k] = mk_;
                L.mag[f][200 - k] = mm_;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();

    // ------------------------------------------------------------------ phases 4 + 5
    // One flat pass over the non-zero 4-bin blocks of all five 16-mel tiles (they are contiguous in
    // plan->fbt), A operands running eight MFMAs ahead in a register ring across tile boundaries;
    // a tile's 16 x 8 outputs are masked and stored when its last block has been accumulated.
    {
        const int fcol = lane & 15, kq = lane >> 4;
        const float* magrow = L.mag[fcol & 7];
        const int t = t0 + (fcol & 7);
        const bool tmask = (t >= Tb) || (t >= mt0 && t < mt0 + mtw);
        const bool store_lane = fcol < kF && t < T_max;
        int tnb[kTiles], tk0[kTiles];
#pragma unroll
        for (int i = 0; i < kTiles; ++i) { tnb[i] = plan->tile_nblk[i]; tk0[i] = plan->tile_k0[i]; }
        const int total = plan->tile_blk0[kTiles - 1] + tnb[kTiles - 1];
        const float* a_img = plan->fbt[0] + lane;
        constexpr int RING = 8;
        float ar[RING];
#pragma unroll
        for (int r = 0; r < RING; ++r) ar[r] = a_img[min(r, max(total - 1, 0)) * 64];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        int tile = 0, left = tnb[0], kbase = tk0[0];
        auto flush = [&]() {
            if (store_lane) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mel = tile * 16 + kq * 4 + r;
                    float v = acc[r];
                    if (flags & 1) v = logf(v + 1e-6f);
                    if (tmask || (mel >= mf0 && mel < mf0 + mfw)) v = 0.f;
                    outb[(int64_t)mel * T_max + t] = v;
                }
            }
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            ++tile;
            // tile metadata by select chain (register arrays must not be indexed at run time)
            int nb = 0, k0n = 0;
#pragma unroll
            for (int i = 1; i < kTiles; ++i) { nb = tile == i ? tnb[i] : nb; k0n = tile == i ? tk0[i] : k0n; }
            left = nb;
            kbase = k0n;
        };
        while (tile < kTiles && left == 0) flush();   // leading all-zero tiles
        for (int g0 = 0; g0 < total; g0 += RING) {
#pragma unroll
            for (int r = 0; r < RING; ++r) {
                const int g = g0 + r;
                const float av = ar[r];
                ar[r] = a_img[min(g + RING, total - 1) * 64];   // eight MFMAs ahead
                if (g < total) {
                    const float bv = magrow[min(kbase + kq, kBins - 1)];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                    kbase += 4;
                    if (--left == 0) {
                        flush();
                        while (tile < kTiles && left == 0) flush();   // all-zero tiles store zeros
                    }
                }
            }
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
    int blk = 0;
    for (int tile = 0; tile < kTiles; ++tile) {
        int lo = kBins, hi = -1;
        for (int f = 0; f < kBins; ++f)
            for (int m = 16 * tile; m < 16 * tile + 16; ++m)
                if (fbank_host[f * kMels + m] != 0.f) {
                    lo = f < lo ? f : lo;
                    hi = f > hi ? f : hi;
                }
        hp.tile_blk0[tile] = blk;
        hp.tile_k0[tile] = hi < 0 ? 0 : lo;
        hp.tile_nblk[tile] = hi < 0 ? 0 : (hi - lo + 4) / 4;
        for (int kb = 0; kb < hp.tile_nblk[tile]; ++kb, ++blk)
            for (int l = 0; l < 64; ++l) {
                const int f = lo + 4 * kb + (l >> 4), m = 16 * tile + (l & 15);
                hp.fbt[blk][l] = f < kBins ? fbank_host[f * kMels + m] : 0.f;
            }
    }
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
