// Conv / BatchNorm / ReLU residual stack of the encoder for gfx950, forward and backward.
// Replaces net.py:9-42 and :56-59 of the reference (cuDNN Conv2d + BatchNorm2d + ReLU, NCHW) with
// channel-last bf16 kernels: activations are [B][F][T][32] (64 bytes per position), so that one
// position's 32 channels are one MFMA K/N fragment row and every access is a 16-byte chunk.
//
//   conv3x5_kernel      3x5 convolution 32->32 as an implicit GEMM on v_mfma_f32_32x32x16_bf16:
//                       M = 32 output channels (A = weights, held in 120 VGPRs for the whole
//                       kernel), N = 32 positions, K = 16 input channels of one tap; 30 MFMAs per
//                       32 positions.  The input strip (6 rows x 68 positions) is staged once
//                       through registers into LDS with the producer's BatchNorm(+ReLU) -- or the
//                       BatchNorm-backward affine map -- applied on the way, so normalised
//                       activations are never materialised.  Same kernel does backward-data
//                       (flipped/transposed weight pack) with a ReLU-mask or residual-add epilogue.
//   conv3x5_wgrad_kernel weight gradient: contraction over positions (K), operands transposed into
//                       LDS ([channel][position]); 5 waves = 5 kw taps, each holds 3 kh
//                       accumulators; a 3-row ring of the activation, 5 shifted copies of dy.
//   stem kernels        1->32 3x3 stride-2 convolution on the f32 spectrogram (VALU; K = 9).
//   bn_* / block_*      per-channel statistics, finalisation to affine coefficients, the block's
//                       residual output z = relu(bn2(y2) + x) and its backward entry.
// LDS rows are padded (80-byte positions, 144-byte channel rows) so every ds_read_b128 group hits
// 16 distinct 16-byte slots.
#include "common.h"

namespace sp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kC = 32;
constexpr int kBnMaxParts = 1024, kBnCountWord = 8 * 64 + kBnMaxParts * 64;   // see SP_BN_SUMS_WORDS

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ uint32_t bf16_bits(float x) {
    return (uint32_t)__builtin_bit_cast(unsigned short, __float2bfloat16(x));
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = bf16_bits(f[0]) | (bf16_bits(f[1]) << 16);
    v.y = bf16_bits(f[2]) | (bf16_bits(f[3]) << 16);
    v.z = bf16_bits(f[4]) | (bf16_bits(f[5]) << 16);
    v.w = bf16_bits(f[6]) | (bf16_bits(f[7]) << 16);
    return v;
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
    uint2 v;
    v.x = bf16_bits(a) | (bf16_bits(b) << 16);
    v.y = bf16_bits(c) | (bf16_bits(d) << 16);
    return v;
}

// shuffle-reduce NV values over the 16 lanes of a wave that share (lane & 3); defined below
template <int NV>
__device__ __forceinline__ void reduce_ch4(float (&v)[NV]);

enum { IN_ID = 0, IN_BNRELU = 1, IN_BNBWD = 2 };
enum { EPI_NONE = 0, EPI_MASK = 1, EPI_ADD = 2 };

// ---------------------------------------------------------------------------------------------
// weight pack: wpack[(tap*2+half)*64 + lane][8], lane (r = lane&31, h = lane>>5), element j =
// Weff[out = r][in = 16 half + 8 h + j][tap].  bwd_data: Weff[ci][co][kh][kw] = w[co][ci][2-kh][4-kw].
// ---------------------------------------------------------------------------------------------
constexpr int kConvPackBatch = 16;
struct ConvPackBatch { const float* w[kConvPackBatch]; __hip_bfloat16* wpack[kConvPackBatch]; };

// blockIdx.y = weight of the batch: every convolution of the stack packed in one launch
__global__ void conv_pack_kernel(ConvPackBatch pb, int bwd_data) {
    const float* __restrict__ w = pb.w[blockIdx.y];
    __hip_bfloat16* __restrict__ wpack = pb.wpack[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (bwd_data == 2) {   // both layouts in one launch: forward in wpack[0:N], backward-data in wpack[N:2N]
        if (i >= 2 * 15 * 2 * 64 * 8) return;
        bwd_data = i >= 15 * 2 * 64 * 8;
        if (bwd_data) { i -= 15 * 2 * 64 * 8; wpack += 15 * 2 * 64 * 8; }
    }
    if (i >= 15 * 2 * 64 * 8) return;
    const int j = i & 7, lane = (i >> 3) & 63, half = (i >> 9) & 1, tap = i >> 10;
    const int r = lane & 31, h = lane >> 5;
    const int kin = 16 * half + 8 * h + j;
    const int kh = tap / 5, kw = tap % 5;
    float v;
    if (!bwd_data) v = w[((r * 32 + kin) * 3 + kh) * 5 + kw];
    else v = w[((kin * 32 + r) * 3 + (2 - kh)) * 5 + (4 - kw)];
    wpack[i] = __float2bfloat16(v);
}

// ---------------------------------------------------------------------------------------------
// conv3x5 (forward / backward-data)
// ---------------------------------------------------------------------------------------------
constexpr int kTP = 64, kFR = 4, kPW = kTP + 4, kRows = kFR + 2, kPStr = 40;  // bf16 elems / position

// Persistent: a workgroup keeps the weights in registers and walks tiles (4 f-rows x 64 positions)
// with stride gridDim.x.  The next tile's input strip (and this tile's epilogue source) is requested
// before this tile's MFMAs, so one HBM round trip is hidden behind a tile of matrix work; per tile
// two barriers (strip written / strip no longer read).
template <int IN_MODE, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x5_kernel(
    const __hip_bfloat16* __restrict__ in, const __hip_bfloat16* __restrict__ in2,
    const float* __restrict__ coef, const __hip_bfloat16* __restrict__ wpack,
    const float* __restrict__ bias, const __hip_bfloat16* __restrict__ epi_src,
    const float* __restrict__ epi_coef, int B, int F, int T, __hip_bfloat16* __restrict__ out,
    float* __restrict__ bn_sums) {
    __shared__ __attribute__((aligned(16))) __hip_bfloat16 tile[kRows * kPW * kPStr];
    __shared__ __attribute__((aligned(16))) __hip_bfloat16 otile[kFR * kTP * kPStr];
    __shared__ __attribute__((aligned(16))) float scoef[5][32];   // input c0 c1 c2 | mask e0 e1
    __shared__ float sbias[32];
    // BatchNorm statistics of the OUTPUT, fused: per-thread running sums live in LDS (the accumulators
    // are free registers only inside the epilogue), [k][thread]: k < 8 first moment, k >= 8 second
    __shared__ float sstat[16][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int ch4 = tid & 3;
    const int ntx = (T + kTP - 1) / kTP, nfy = (F + kFR - 1) / kFR;
    const int ntiles = ntx * nfy * B;

    if (tid < 96) scoef[tid >> 5][tid & 31] = (IN_MODE != IN_ID && (IN_MODE == IN_BNBWD || tid < 64)) ? coef[tid] : 0.f;
    else if (tid < 160) scoef[3 + ((tid - 96) >> 5)][tid & 31] = (EPI == EPI_MASK) ? epi_coef[tid - 96] : 0.f;
    // ---- weights: the whole 32x480 matrix in registers (30 fragments), once per workgroup ----
    bf16x8 wreg[30];
#pragma unroll
    for (int i = 0; i < 30; ++i)
        wreg[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wpack + ((int64_t)i * 64 + lane) * 8));
    if (tid < 32) sbias[tid] = (EPI == EPI_NONE && bias) ? bias[tid] : 0.f;
    const bool stats = bn_sums != nullptr && (EPI == EPI_NONE || EPI == EPI_MASK);
#pragma unroll
    for (int k = 0; k < 16; ++k) sstat[k][tid] = 0.f;

    constexpr int NIT = (kRows * kPW * 4 + 255) / 256;
    uint4 v[NIT], v2[NIT];
    bool ok[NIT];
    auto decode = [&](int tl, int& t0, int& f0, int64_t& img) {
        const int tx = tl % ntx, rest = tl / ntx;
        t0 = tx * kTP;
        f0 = (rest % nfy) * kFR;
        img = (int64_t)(rest / nfy) * F * T;
    };
    auto load_strip = [&](int tl) {   // global -> registers, nothing waits on it here
        // branch-free: every lane loads from a clamped (valid) address and `ok` decides later whether the value
        // or the zero padding is stored -- an exec-mask branch per guarded load cost ~30 cycles apiece
        int t0, f0;
        int64_t img;
        decode(min(tl, ntiles - 1), t0, f0, img);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = min(tid + it * 256, kRows * kPW * 4 - 1);
            const int row = id / (kPW * 4);
            const int pos = (id - row * (kPW * 4)) >> 2;
            const int f = f0 - 1 + row, t = t0 - 2 + pos;
            ok[it] = tl < ntiles && f >= 0 && f < F && t >= 0 && t < T;
            const int64_t off = ((img + (int64_t)min(max(f, 0), F - 1) * T + min(max(t, 0), T - 1)) * kC + ch4 * 8);
            v[it] = *reinterpret_cast<const uint4*>(in + off);
            if (IN_MODE == IN_BNBWD) v2[it] = *reinterpret_cast<const uint4*>(in2 + off);
            else v2[it] = make_uint4(0, 0, 0, 0);
        }
    };
    auto store_strip = [&]() {   // registers -> LDS with the producer's transform applied
        const float* c0 = scoef[0] + ch4 * 8;
        const float* c1 = scoef[1] + ch4 * 8;
        const float* c2 = scoef[2] + ch4 * 8;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int id = tid + it * 256;
            const int row = id / (kPW * 4);
            const int pos = (id - row * (kPW * 4)) >> 2;
            uint4 o = v[it];
            if (IN_MODE == IN_BNRELU) {
                float x[8];
                unpack8(o, x);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = fmaxf(0.f, c0[j] * x[j] + c1[j]);
                o = pack8(x);
            } else if (IN_MODE == IN_BNBWD) {
                float x[8], y[8];
                unpack8(o, x);
                unpack8(v2[it], y);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = c0[j] * x[j] + c1[j] * y[j] + c2[j];
                o = pack8(x);
            }
            if (!ok[it]) o = make_uint4(0, 0, 0, 0);   // zero padding is applied AFTER the transform
            if (id < kRows * kPW * 4) *reinterpret_cast<uint4*>(tile + (row * kPW + pos) * kPStr + ch4 * 8) = o;
        }
    };

    // The BN-backward strip is two tensors (56 VGPRs): held across the MFMAs it pushes the kernel past
    // 256 registers, so only the single-tensor modes prefetch the next tile's strip.
    constexpr bool kPrefetch = IN_MODE != IN_BNBWD;
    // running BatchNorm sums: registers where there is room (forward modes), LDS otherwise
    constexpr bool kStatRegs = EPI == EPI_NONE;
    float rs[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) rs[k] = 0.f;
    __syncthreads();   // scoef
    int tl = blockIdx.x;
    if (kPrefetch) load_strip(tl);
    for (; tl < ntiles; tl += gridDim.x) {
        int t0, f0;
        int64_t img;
        decode(tl, t0, f0, img);
        if (!kPrefetch) load_strip(tl);
        store_strip();
        __syncthreads();                 // strip of this tile is in LDS
        if (kPrefetch) load_strip(tl + gridDim.x);   // next tile's strip: lands during the MFMAs below
        const int f = f0 + wave;
        // epilogue source of this tile, requested before the matrix work
        uint4 ev[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int pos = it * 16 + (lane >> 2), t = t0 + pos;
            ev[it] = make_uint4(0, 0, 0, 0);
            if (EPI != EPI_NONE)    // clamped address: rows / positions past the edge are never stored
                ev[it] = *reinterpret_cast<const uint4*>(epi_src + (img + (int64_t)min(f, F - 1) * T + min(t, T - 1)) * kC + ch4 * 8);
        }
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[m][q] = sbias[8 * (q >> 2) + 4 * h + (q & 3)];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const __hip_bfloat16* p = tile + ((wave + kh) * kPW + m * 32 + r + kw) * kPStr + (half * 2 + h) * 8;
                        const bf16x8 bfrag = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p));
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[(kh * 5 + kw) * 2 + half], bfrag, acc[m], 0, 0, 0);
                    }
                }
            }
        }
        // ---- epilogue ----
        // The MFMA leaves lane = position, registers = 16 of its channels (8g + 4h + i): stored directly
        // that is 8 bytes per lane into 32 different cache lines per instruction.  Each wave therefore
        // transposes its 64 x 32 output through its own LDS slab and then works in 16-byte chunks
        // (4 lanes = one position's 64 bytes, consecutive positions in consecutive lanes).
        if (f < F) {
            __hip_bfloat16* slab = otile + wave * (kTP * kPStr);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<uint2*>(slab + (m * 32 + r) * kPStr + 8 * g + 4 * h) =
                        pack4(acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]);
            }
            __builtin_amdgcn_wave_barrier();
            const float* e0 = scoef[3] + ch4 * 8;
            const float* e1 = scoef[4] + ch4 * 8;
            float ps[8], pq[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { ps[j] = 0.f; pq[j] = 0.f; }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pos = it * 16 + (lane >> 2), t = t0 + pos;
                uint4 o = *reinterpret_cast<const uint4*>(slab + pos * kPStr + ch4 * 8);
                float e[8];
                if (EPI != EPI_NONE) {
                    float x[8];
                    unpack8(o, x);
                    unpack8(ev[it], e);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (EPI == EPI_MASK) x[j] = (e0[j] * e[j] + e1[j] > 0.f) ? x[j] : 0.f;
                        else x[j] += e[j];
                    }
                    o = pack8(x);
                }
                if (t < T) {
                    *reinterpret_cast<uint4*>(out + (img + (int64_t)f * T + t) * kC + ch4 * 8) = o;
                    if (stats) {   // on the values as stored (bf16): what a separate reduction pass would read
                        float x[8];
                        unpack8(o, x);
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            ps[j] += x[j];
                            pq[j] += (EPI == EPI_MASK) ? x[j] * e[j] : x[j] * x[j];   // (sum g, sum g*y) | (sum y, sum y^2)
                        }
                    }
                }
            }
            if (stats) {
                if (kStatRegs) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { rs[j] += ps[j]; rs[8 + j] += pq[j]; }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        sstat[j][tid] += ps[j];
                        sstat[8 + j][tid] += pq[j];
                    }
                }
            }
        }
        __syncthreads();                 // nobody reads this tile's strip any more
    }
    if (stats) {   // workgroup reduction over the threads that share a channel group, one 64-float partial per workgroup
        float a16[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) a16[k] = kStatRegs ? rs[k] : sstat[k][tid];
        reduce_ch4<16>(a16);
        __syncthreads();
        float* red = &sstat[0][0];   // reuse as [4 waves][4 groups][16]
        if (lane < 4) {
#pragma unroll
            for (int k = 0; k < 16; ++k) red[(wave * 4 + lane) * 16 + k] = a16[k];
        }
        __syncthreads();
        if (tid < 64) {
            const int c4 = tid >> 4, k = tid & 15;
            const float sm = red[(0 * 4 + c4) * 16 + k] + red[(1 * 4 + c4) * 16 + k] + red[(2 * 4 + c4) * 16 + k] +
                             red[(3 * 4 + c4) * 16 + k];
            bn_sums[(8 + blockIdx.x) * 64 + (k >> 3) * 32 + c4 * 8 + (k & 7)] = sm;   // partial; see bn_rows
        }
        if (blockIdx.x == 0 && tid == 0) bn_sums[kBnCountWord] = __int_as_float((int)gridDim.x);
    }
}

// ---------------------------------------------------------------------------------------------
// conv3x5 weight gradient
// ---------------------------------------------------------------------------------------------
// Operands: D[co][ci] (per tap) += sum_pos dy[pos][co] * a[pos + tap][ci], K = positions.  The MFMA wants 8
// consecutive k (positions) per lane while memory holds consecutive channels per position.  The rows are staged
// in LDS exactly as they lie in memory -- [position][32 channels], one 16-byte store per thread, the producers'
// BatchNorm transforms applied in registers on the way -- and the fragments are read with ds_read_b64_tr_b16,
// which hands each lane one COLUMN (channel) of a 4 (positions) x 16 (channels) block: two reads = 8 positions.
// A tap's kw shift is an address offset of whole 64-byte rows, so all five taps read the same image.
// (The first version transposed while staging -- sixteen 16-bit LDS stores per thread and row -- and shifted
// 32-byte windows in registers: 190 vector instructions per row and wave against 12 MFMAs.)
//
// A workgroup is kWG independent groups of five waves (one (image, row range, 64-position chunk) item
// each; wave = kw tap, three kh accumulators) that meet again only at the end: the groups' accumulators are
// added through LDS and ONE of them issues the float atomics.  The atomics are what this buys: every workgroup
// result is 15 x 32 x 32 dword atomics, the L2 executes them one dword at a time, and with one item per
// workgroup they were a third of the kernel's time.
constexpr int kWG = 3;
typedef short wg_s16x4 __attribute__((ext_vector_type(4)));
typedef short wg_s16x8 __attribute__((ext_vector_type(8)));

// 8 consecutive positions (rows of 32 channels = 64 bytes) of this lane's channel, from LDS byte address `addr`
// (the lane's row 8 kg + qq, 8-byte piece p of channel half hm: see the caller)
__device__ __forceinline__ bf16x8 wgrad_frag(uint32_t addr) {
    typedef __attribute__((address_space(3))) wg_s16x4* lp;
    const wg_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)addr);
    const wg_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(uintptr_t)(addr + 4 * 64));
    const wg_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int A_MODE>
__global__ __launch_bounds__(320 * kWG) void conv3x5_wgrad_kernel(
    const __hip_bfloat16* __restrict__ asrc, const float* __restrict__ acoef,
    const __hip_bfloat16* __restrict__ g, const __hip_bfloat16* __restrict__ y,
    const float* __restrict__ gcoef, int B, int F, int T, int fsplit, int ntc, float* __restrict__ gwp) {
    // per group: four activation rows [64 positions][32] (rows f-1, f, f+1 in use, f+2 being written) and two
    // dy rows [68 positions, origin t0-2][32] (row f in use, f+1 being written): one barrier per row
    constexpr int kARow = 64 * 32, kGRow = 68 * 32, kASlots = 4, kGSlots = 2;
    constexpr int kGrp = kASlots * kARow + kGSlots * kGRow;          // bf16 elements per group (24.5 KB)
    __shared__ __attribute__((aligned(256))) __hip_bfloat16 smem[kWG * kGrp];
    static_assert(sizeof(__hip_bfloat16) * kWG * kGrp >= sizeof(float) * (kWG - 1) * 5 * 16 * 64,
                  "the staging buffers double as the end-of-kernel reduction buffer");
    const int grp = threadIdx.x / 320, tid = threadIdx.x - grp * 320;   // 320 = 5 waves: groups are wave-aligned
    __hip_bfloat16* aring = smem + grp * kGrp;
    __hip_bfloat16* gring = aring + kASlots * kARow;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // item = (image b, row range fs, position chunk tc); a group past the last item runs empty rows.  The
    // workgroups are persistent: group (blockIdx.x, grp) takes items blockIdx.x * kWG + grp + k * gridDim.x * kWG
    // and keeps accumulating in the same registers, so there is one result per workgroup however large the batch
    const int nitems = ntc * B * fsplit, fper = (F + fsplit - 1) / fsplit;
    const int niter = (nitems + (int)gridDim.x * kWG - 1) / ((int)gridDim.x * kWG);
    bool live = false;
    int t0 = 0, fa = 0, fb = 0;
    int64_t img = 0;
    const int ch4 = tid & 3;

    // per-channel coefficients live in LDS (40 VGPRs otherwise)
    __shared__ __attribute__((aligned(16))) float scoef[5][32];   // ga gb gd a0 a1
    if (threadIdx.x < 96) scoef[tid >> 5][tid & 31] = gcoef[tid];
    else if (threadIdx.x < 160) scoef[3 + ((tid - 96) >> 5)][tid & 31] = (A_MODE == IN_BNRELU) ? acoef[tid - 96] : ((tid < 128) ? 1.f : 0.f);
    __syncthreads();
    const float* ga = scoef[0] + ch4 * 8;
    const float* gb = scoef[1] + ch4 * 8;
    const float* gd = scoef[2] + ch4 * 8;
    const float* a0 = scoef[3] + ch4 * 8;
    const float* a1 = scoef[4] + ch4 * 8;

    // global -> registers (issued PF rows ahead) and registers -> LDS are split so that a row's HBM latency
    // hides behind the previous rows' MFMAs
    struct RowRegs { uint4 a, g, y; bool aok, gok; };
    auto load_row = [&](int frow_a, int frow_g, RowRegs& rr) {
        rr.a = rr.g = rr.y = make_uint4(0, 0, 0, 0);
        const int pos = tid >> 2, ta = t0 + pos;
        rr.aok = live && tid < 256 && frow_a >= 0 && frow_a < F && ta < T;
        if (rr.aok) rr.a = *reinterpret_cast<const uint4*>(asrc + (img + (int64_t)frow_a * T + ta) * kC + ch4 * 8);
        const int tg = t0 - 2 + pos;
        rr.gok = live && tid < 272 && frow_g >= 0 && frow_g < F && tg >= 0 && tg < T;
        if (rr.gok) {
            const int64_t off = (img + (int64_t)frow_g * T + tg) * kC + ch4 * 8;
            rr.g = *reinterpret_cast<const uint4*>(g + off);
            rr.y = *reinterpret_cast<const uint4*>(y + off);
        }
    };
    auto store_a = [&](int frow, const RowRegs& rr) {   // activation row -> slot frow mod 4, [pos][ci]
        if (tid < 256) {
            uint4 o = rr.a;                              // zeros where the row or the position is padding
            if (A_MODE == IN_BNRELU) {
                float x[8];
                unpack8(rr.a, x);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = rr.aok ? fmaxf(0.f, a0[j] * x[j] + a1[j]) : 0.f;
                o = pack8(x);
            }
            *reinterpret_cast<uint4*>(aring + ((frow + kASlots) % kASlots) * kARow + tid * 8) = o;
        }
    };
    auto store_g = [&](int frow, const RowRegs& rr) {   // dy row (BN-backward affine applied) -> slot frow mod 2, [idx][co]
        if (tid < 272) {
            float x[8], yv[8];
            unpack8(rr.g, x);
            unpack8(rr.y, yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = rr.gok ? ga[j] * x[j] + gb[j] * yv[j] + gd[j] : 0.f;
            *reinterpret_cast<uint4*>(gring + (frow & 1) * kGRow + tid * 8) = pack8(x);
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[k][q] = 0.f;

    // fragment address of this lane: a 16-lane group reads a 4 (positions) x 16 (channels) block, lane (qq, p)
    // supplies row qq's 8-byte piece p and receives column 4 qq + p; hm = channel half, kg = k group of the MFMA
    const int li = lane & 15, qq = li >> 2, pp = li & 3, hm = (lane >> 4) & 1, kg = lane >> 5;
    const uint32_t lane_off = (uint32_t)((8 * kg + qq) * 64 + hm * 32 + pp * 8);
    typedef __attribute__((address_space(3))) void* lds_vp;
    const uint32_t a_base = (uint32_t)(uintptr_t)(lds_vp)aring + lane_off;
    // tap kw = wave reads dy at row offset 4 - kw of the 68-position window
    const uint32_t g_base = (uint32_t)(uintptr_t)(lds_vp)gring + lane_off + (uint32_t)(4 - wave) * 64;

    constexpr int PF = 2;      // rows in flight per thread; 3 pushes the kernel past 128 VGPRs (four waves per SIMD are needed for the 15-wave workgroup): 60 -> 68 us
    for (int iter = 0; iter < niter; ++iter) {
        const int item = (iter * (int)gridDim.x + (int)blockIdx.x) * kWG + grp;
        live = item < nitems;
        const int tc = item % ntc, bf = min(item / ntc, B * fsplit - 1);
        t0 = tc * 64;
        const int b = bf / fsplit, fs = bf % fsplit;
        fa = fs * fper;
        fb = live ? min(F, fa + fper) : fa;
        img = (int64_t)b * F * T;
        RowRegs q[PF];
        {
            RowRegs p0, p1, p2;
            load_row(fa - 1, -1, p0);
            load_row(fa, -1, p1);
            load_row(fa + 1, fa < fb ? fa : -1, p2);
#pragma unroll
            for (int i = 0; i < PF; ++i) load_row(fa + 2 + i, fa + 1 + i < fb ? fa + 1 + i : -1, q[i]);
            store_a(fa - 1, p0);
            store_a(fa, p1);
            store_a(fa + 1, p2);
            store_g(fa, p2);
        }
        __syncthreads();
        for (int f = fa; f < fa + fper; ++f) {   // every group runs fper rows (the barriers are the workgroup's): rows past fb are zeros
            // rows of the NEXT iteration: their slots were last read in the previous iteration
            store_a(f + 2, q[0]);
            store_g(f + 1, q[0]);
#pragma unroll
            for (int i = 0; i + 1 < PF; ++i) q[i] = q[i + 1];
            load_row(f + 2 + PF, f + 1 + PF < fb ? f + 1 + PF : -1, q[PF - 1]);
            const uint32_t gs = g_base + (uint32_t)((f & 1) * kGRow * 2);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 A = wgrad_frag(gs + ks * 16 * 64);
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int slot = (f + kh - 1 + kASlots) % kASlots;
                    const bf16x8 Bv = wgrad_frag(a_base + (uint32_t)(slot * kARow * 2 + ks * 16 * 64));
                    acc[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, acc[kh], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    // D[row = co][col = ci]: lane = ci, registers = co 8g + 4h + i; gwp[workgroup][tap][co][ci].
    // Groups 1 .. kWG-1 hand their accumulators to group 0 through LDS (the staging buffers are free:
    // the loop ended in a barrier), one kh plane at a time, added in a fixed order, and group 0 STORES the
    // workgroup's sum to the workgroup's own slab: no float atomics (the L2 executes those one dword at a time --
    // 15 360 per workgroup, all arriving when the single round of workgroups ends), and the unpack kernel's
    // fixed-order sum over the slabs makes the weight gradient bit-reproducible.
    float* rbuf = reinterpret_cast<float*>(smem);   // [kWG-1][5 waves][16][64]
    float* slab = gwp + (size_t)blockIdx.x * (15 * 32 * 32);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        if (grp > 0) {
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) rbuf[(((grp - 1) * 5 + wave) * 16 + q2) * 64 + lane] = acc[kh][q2];
        }
        __syncthreads();
        if (grp == 0) {
            const int tap = kh * 5 + wave;
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) {
                float v = acc[kh][q2];
#pragma unroll
                for (int o = 0; o < kWG - 1; ++o) v += rbuf[((o * 5 + wave) * 16 + q2) * 64 + lane];
                const int co = 8 * (q2 >> 2) + 4 * h + (q2 & 3);
                slab[(tap * 32 + co) * 32 + r] = v;
            }
        }
        __syncthreads();
    }
}

// gw[co][ci][kh][kw] += sum over the workgroups' slabs, in a fixed order.  A workgroup of 16 waves owns 64
// consecutive outputs; wave w adds slabs w, w + 16, ... (every load of a thread is in flight before the first
// add: the slabs come from beyond the L2 and a second dependent batch would double the kernel's time), then the
// sixteen partial sums are added through LDS in a fixed order.
__global__ __launch_bounds__(1024) void conv_wgrad_unpack_kernel(const float* __restrict__ gwp, int nslabs, float* __restrict__ gw) {
    constexpr int NL = SP_CONV_WGRAD_SLABS / 16;
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;                 // over [tap][co][ci]: a wave reads 256 contiguous bytes of a slab
    float v[NL];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int c = w + 16 * j;
        v[j] = c < nslabs ? gwp[(size_t)c * (15 * 32 * 32) + i] : 0.f;
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NL; ++j) sum += v[j];
    part[w][lane] = sum;
    __syncthreads();
    if (w == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][lane];
        const int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
        gw[(co * 32 + ci) * 15 + tap] += t;
    }
}

// ---------------------------------------------------------------------------------------------
// stem: Conv2d(1, 32, 3, stride 2, pad 1) on the f32 spectrogram [B][Fin][Tin]
// ---------------------------------------------------------------------------------------------
// Thread = (position t, channel group ch4 = tid & 3).  The four lanes of a position share its input
// taps: each lane loads one tap column (branch-free: clamped index, zeroed by a select when it falls
// in the padding) and the quad exchanges them with DPP broadcasts.
constexpr int kStemRows = 4;   // output rows per forward work item

template <int K>
__device__ __forceinline__ float quad_bcast(float v) {   // lane K of every aligned group of four
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}

__device__ __forceinline__ float stem_tap(const float* __restrict__ sb, int fi, int Fin, int Tin, int tic, bool tok) {
    const bool rowok = fi >= 0 && fi < Fin;
    const float v = sb[(int64_t)min(max(fi, 0), Fin - 1) * Tin + tic];
    return (rowok && tok) ? v : 0.f;
}

// The 72 weights and 8 biases of a thread's 8 output channels stay in registers; the workgroup is
// persistent over work items = (utterance, kStemRows output rows, 64 positions) and requests the next
// item's input taps before it computes the current one; a wave's store writes 1 KB of contiguous output.
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ spec,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ bias, int B, int Fin,
                                                       int Tin, int F, int T,
                                                       __hip_bfloat16* __restrict__ out) {
    const int ch4 = threadIdx.x & 3, tl = threadIdx.x >> 2, kw = min(ch4, 2);
    float wr[72], bs[8];   // w is [co][9]: this thread's 8 channels are 72 contiguous floats
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(w + ch4 * 72 + i * 4);
        wr[i * 4] = v.x; wr[i * 4 + 1] = v.y; wr[i * 4 + 2] = v.z; wr[i * 4 + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = bias[ch4 * 8 + j];
    const int fgroups = (F + kStemRows - 1) / kStemRows, nch = (T + 63) / 64;
    const int items = B * fgroups * nch;
    constexpr int NR = 2 * kStemRows + 1;   // input rows under kStemRows output rows
    auto load = [&](int it, float (&v)[NR]) {
        const int bf = it / nch, tc = it - bf * nch;
        const int b = bf / fgroups, f0 = (bf - b * fgroups) * kStemRows;
        const int ti = 2 * min(tc * 64 + tl, T - 1) - 1 + kw;
        const bool tok = ti >= 0 && ti < Tin;
        const int tic = min(max(ti, 0), Tin - 1);
        const float* sb = spec + (int64_t)b * Fin * Tin;
#pragma unroll
        for (int i = 0; i < NR; ++i) v[i] = stem_tap(sb, 2 * f0 - 1 + i, Fin, Tin, tic, tok);
    };
    float cur[NR], nxt[NR];
    int it = blockIdx.x;
    if (it < items) load(it, cur);
    for (; it < items; it += gridDim.x) {
        const int itn = it + gridDim.x;
        if (itn < items) load(itn, nxt);
        const int bf = it / nch, tc = it - bf * nch;
        const int b = bf / fgroups, f0 = (bf - b * fgroups) * kStemRows;
        const int t = tc * 64 + tl;
        float x[NR][3];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            x[i][0] = quad_bcast<0>(cur[i]);
            x[i][1] = quad_bcast<1>(cur[i]);
            x[i][2] = quad_bcast<2>(cur[i]);
        }
#pragma unroll
        for (int fr = 0; fr < kStemRows; ++fr) {
            const int f = f0 + fr;
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = bs[j];
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) o[j] += wr[j * 9 + tap] * x[2 * fr + tap / 3][tap % 3];
            }
            if (f < F && t < T)
                *reinterpret_cast<uint4*>(out + (((int64_t)b * F + f) * T + t) * kC + ch4 * 8) = pack8(o);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) cur[i] = nxt[i];
    }
}

// shuffle-reduce NV values over the 16 lanes of a wave that share (lane & 3)
template <int NV>
__device__ __forceinline__ void reduce_ch4(float (&v)[NV]) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1)
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += __shfl_xor(v[i], o);
}

// Each thread owns one 16-byte chunk (8 channels, group tid & 3) of a position and 80 accumulators.
// Work items = (output row (b, f), chunk of 64 positions), walked by groups of four waves; a
// workgroup holds kStemBwdGroups of them so that few workgroups finish in the 320 float atomics that
// end the kernel (they serialise per address: ~20 ns each).  The loop is software-pipelined: the
// loads of the next two items are requested before the current two are multiplied; lane ch4 of a
// position loads input row min(ch4, 2) and the quad exchanges the taps with DPP broadcasts.
constexpr int kStemBwdGroups = 3;

__global__ __launch_bounds__(256 * kStemBwdGroups) void stem_bwd_kernel(const float* __restrict__ spec,
                                                       const __hip_bfloat16* __restrict__ gout, int B,
                                                       int Fin, int Tin, int F, int T,
                                                       float* __restrict__ gw, float* __restrict__ gbias) {
    constexpr int NW = 4 * kStemBwdGroups;
    __shared__ float red[NW][4][80];
    const int ch4 = threadIdx.x & 3, tl = (threadIdx.x & 255) >> 2, kh = min(ch4, 2);
    const int group = blockIdx.x * kStemBwdGroups + (threadIdx.x >> 8), ngroups = gridDim.x * kStemBwdGroups;
    float acc[80];
#pragma unroll
    for (int i = 0; i < 80; ++i) acc[i] = 0.f;
    const int nch = (T + 63) / 64, items = B * F * nch;
    auto load = [&](int it, uint4& gr, float (&v)[3]) {
        const int itc = min(it, items - 1);
        const int row = itc / nch, tt = (itc - row * nch) * 64 + tl;
        const bool ok = it < items && tt < T;
        const int t = min(tt, T - 1);
        const int b = row / F, f = row - b * F;
        const float* sb = spec + (int64_t)b * Fin * Tin;
        gr = *reinterpret_cast<const uint4*>(gout + ((int64_t)row * T + t) * kC + ch4 * 8);
        if (!ok) gr = make_uint4(0, 0, 0, 0);   // past the end: zero gradient
        const int ti0 = 2 * t - 1;
        v[0] = stem_tap(sb, 2 * f - 1 + kh, Fin, Tin, max(ti0, 0), ti0 >= 0);
        v[1] = stem_tap(sb, 2 * f - 1 + kh, Fin, Tin, ti0 + 1, true);               // 2 t <= Tin - 1 always
        v[2] = stem_tap(sb, 2 * f - 1 + kh, Fin, Tin, min(ti0 + 2, Tin - 1), ti0 + 2 < Tin);
    };
    auto accum = [&](const uint4& gr, const float (&v)[3]) {
        float gch[8], x[9];
        unpack8(gr, gch);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            x[k] = quad_bcast<0>(v[k]);
            x[3 + k] = quad_bcast<1>(v[k]);
            x[6 + k] = quad_bcast<2>(v[k]);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j * 9 + tap] += gch[j] * x[tap];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[72 + j] += gch[j];
    };
    constexpr int NI = 2;
    uint4 gr[NI], gn[NI];
    float v[NI][3], vn[NI][3];
    int it = group;
#pragma unroll
    for (int u = 0; u < NI; ++u) load(it + u * ngroups, gr[u], v[u]);
    for (; it < items; it += NI * ngroups) {
#pragma unroll
        for (int u = 0; u < NI; ++u) load(it + (NI + u) * ngroups, gn[u], vn[u]);
#pragma unroll
        for (int u = 0; u < NI; ++u) accum(gr[u], v[u]);
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            gr[u] = gn[u];
#pragma unroll
            for (int k = 0; k < 3; ++k) v[u][k] = vn[u][k];
        }
    }
    reduce_ch4<80>(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 80; ++i) red[wave][lane][i] = acc[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 320; i += blockDim.x) {
        const int c4 = i / 80, k = i % 80;
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) s += red[wv][c4][k];
        if (k < 72) atomicAdd(gw + (c4 * 8 + k / 9) * 9 + k % 9, s);
        else atomicAdd(gbias + c4 * 8 + (k - 72), s);
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics / finalisation, block output and backward entry
// ---------------------------------------------------------------------------------------------
// sums[0..31] += sum_pos a*b', sums[32..63] += sum_pos second moment, per channel:
//   MODE 0 (forward stats):   (sum y, sum y^2)
//   MODE 1 (backward reduce): (sum g, sum g*y)
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const __hip_bfloat16* __restrict__ p0,
                                                        const __hip_bfloat16* __restrict__ p1, int64_t n,
                                                        float* __restrict__ sums) {
    __shared__ float red[4][4][16];
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // four independent 16-byte loads in flight per thread; the grid is small (reduce_blocks): every
    // block ends in one 64-float partial
    const int64_t stride = (int64_t)gridDim.x * 256, total = n * 4;
    for (int64_t c0 = (int64_t)blockIdx.x * 256 + threadIdx.x; c0 < total; c0 += 4 * stride) {
        uint4 ra[4], rb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t c = c0 + u * stride;
            ra[u] = make_uint4(0, 0, 0, 0);
            rb[u] = make_uint4(0, 0, 0, 0);
            if (c < total) {
                ra[u] = *reinterpret_cast<const uint4*>(p0 + c * 8);
                if (MODE == 1) rb[u] = *reinterpret_cast<const uint4*>(p1 + c * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float a[8], bv[8];
            unpack8(ra[u], a);
            if (MODE == 1) unpack8(rb[u], bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] += a[j];
                acc[8 + j] += (MODE == 0) ? a[j] * a[j] : a[j] * bv[j];
            }
        }
    }
    reduce_ch4<16>(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][lane][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c4 = threadIdx.x >> 4, k = threadIdx.x & 15;
        const float s = red[0][c4][k] + red[1][c4][k] + red[2][c4][k] + red[3][c4][k];
        sums[(8 + blockIdx.x) * 64 + (k >> 3) * 32 + c4 * 8 + (k & 7)] = s;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[kBnCountWord] = __int_as_float((int)gridDim.x);
}

// Second level of the BatchNorm reductions, fused into the finalising kernels.  The producing launch leaves one
// 64-float partial per workgroup at sums[kBnL2 * 64 + 64 b ..] and the number of partials in sums[kBnCountWord].
// The finaliser -- one workgroup of 1024 threads -- adds, for each of eight rows g, the partials b = g, g + 8, ...
// in a FIXED order (four rows at a time, 256 threads per row), keeps the eight row sums in LDS, writes them to
// sums[0 .. 512) for inspection, and adds the rows in a fixed order.  No float atomics anywhere: the statistics,
// and through them every bf16 activation of the stack, are bit-reproducible from run to run and identical on
// data-parallel replicas.  (The second level used to be a launch of its own, eight workgroups: 7 us, twelve
// times a step, for at most 256 KB.)
constexpr int kBnL2 = 8;
__device__ __forceinline__ void bn_rows(float* __restrict__ sums, float (*rows)[64], float (*grp)[4][64]) {
    const int nparts = min(max(__float_as_int(sums[kBnCountWord]), 0), kBnMaxParts);   // 8 * (3 + 96 + 28) + 7 = 1023 is the last index covered
    const int idx = threadIdx.x & 63, q = (threadIdx.x >> 6) & 3, vg = threadIdx.x >> 8;
    const float* part = sums + kBnL2 * 64;
    // every load of this thread is requested before the first add: the partials were just written by other
    // XCDs and come from beyond the L2, so a dependent second batch would double the kernel's 3 us
    float v[2][4][8];
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int g = round * 4 + vg;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int b = g + kBnL2 * (q + 32 * it + 4 * j);     // row g's partials: b = g + 8 i
                v[round][it][j] = b < nparts ? part[b * 64 + idx] : 0.f;
            }
        }
    }
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int g = round * 4 + vg;
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = ((v[round][0][j] + v[round][1][j]) + v[round][2][j]) + v[round][3][j];
        grp[vg][q][idx] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        __syncthreads();
        if (q == 0) {
            const float r = (grp[vg][0][idx] + grp[vg][1][idx]) + (grp[vg][2][idx] + grp[vg][3][idx]);
            rows[g][idx] = r;
            sums[g * 64 + idx] = r;
        }
        __syncthreads();
    }
}

// the reduced sum: the eight second-level rows in a fixed order
__device__ __forceinline__ float bn_sum8(const float (*rows)[64], int i) {
    return ((rows[0][i] + rows[1][i]) + (rows[2][i] + rows[3][i])) + ((rows[4][i] + rows[5][i]) + (rows[6][i] + rows[7][i]));
}

// coef[0..31] = scale, coef[32..63] = shift; save[0..31] = mean, save[32..63] = invstd
__global__ __launch_bounds__(1024) void bn_finalize_fwd_kernel(float* __restrict__ sums, float n, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ rmean,
                                       float* __restrict__ rvar, float momentum, float eps, int training,
                                       float* __restrict__ coef, float* __restrict__ save) {
    __shared__ float rows[kBnL2][64];
    __shared__ float grp[4][4][64];
    if (training) bn_rows(sums, rows, grp);
    const int c = threadIdx.x;
    if (c >= 32) return;
    float mean, var;
    if (training) {
        mean = bn_sum8(rows, c) / n;
        var = fmaxf(bn_sum8(rows, 32 + c) / n - mean * mean, 0.f);
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * var * (n / fmaxf(n - 1.f, 1.f));
    } else {
        mean = rmean[c];
        var = rvar[c];
    }
    const float invstd = rsqrtf(var + eps);
    const float s = gamma[c] * invstd;
    coef[c] = s;
    coef[32 + c] = beta[c] - mean * s;
    save[c] = mean;
    save[32 + c] = invstd;
}

// BN backward: dy = a*g + b*y + d with a = gamma*invstd, b = -a*invstd*dgamma/n,
// d = -a*sum_g/n - b*mean;  dgamma = invstd*(sum_gy - mean*sum_g), dbeta = sum_g.
__global__ __launch_bounds__(1024) void bn_finalize_bwd_kernel(float* __restrict__ sums, const float* __restrict__ save,
                                       const float* __restrict__ gamma, float n, float* __restrict__ coef3,
                                       float* __restrict__ ggamma, float* __restrict__ gbeta) {
    __shared__ float rows[kBnL2][64];
    __shared__ float grp[4][4][64];
    bn_rows(sums, rows, grp);
    const int c = threadIdx.x;
    if (c >= 32) return;
    const float mean = save[c], invstd = save[32 + c];
    const float sg = bn_sum8(rows, c), sgy = bn_sum8(rows, 32 + c);
    const float dgamma = invstd * (sgy - mean * sg);
    const float a = gamma[c] * invstd;
    const float bq = -a * invstd * dgamma / n;
    coef3[c] = a;
    coef3[32 + c] = bq;
    coef3[64 + c] = -a * sg / n - bq * mean;
    ggamma[c] += dgamma;
    gbeta[c] += sg;
}

struct Strides { int64_t sb, sf, st; };

// z = relu(s*y2 + t + x); z may live in another layout (granule address b*sb + f*sf + t*st)
__global__ __launch_bounds__(256) void block_out_kernel(const __hip_bfloat16* __restrict__ y2,
                                                        const float* __restrict__ coef,
                                                        const __hip_bfloat16* __restrict__ x, int B, int F,
                                                        int T, __hip_bfloat16* __restrict__ z, Strides zs) {
    const int ch4 = threadIdx.x & 3;
    float s[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] = coef[ch4 * 8 + j]; sh[j] = coef[32 + ch4 * 8 + j]; }
    const int64_t n = (int64_t)B * F * T;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < n * 4; c += (int64_t)gridDim.x * 256) {
        const int64_t p = c >> 2;
        float a[8], xv[8];
        unpack8(*reinterpret_cast<const uint4*>(y2 + c * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(x + c * 8), xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = fmaxf(0.f, s[j] * a[j] + sh[j] + xv[j]);
        const int t = (int)(p % T);
        const int64_t bf = p / T;
        const int64_t off = (bf / F) * zs.sb + (bf % F) * zs.sf + (int64_t)t * zs.st + ch4 * 8;
        *reinterpret_cast<uint4*>(z + off) = pack8(a);
    }
}

// g2pre = gz * [z > 0] (gz, z in the strided layout), sums += (sum g2pre, sum g2pre*y2)
template <bool DENSE>
__global__ __launch_bounds__(256) void block_bwd_in_kernel(const __hip_bfloat16* __restrict__ gz,
                                                           const __hip_bfloat16* __restrict__ z, Strides zs,
                                                           const __hip_bfloat16* __restrict__ y2, int B, int F,
                                                           int T, __hip_bfloat16* __restrict__ gpre,
                                                           float* __restrict__ sums) {
    __shared__ float red[4][4][16];
    const int ch4 = threadIdx.x & 3;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const uint32_t total = (uint32_t)B * F * T * 4;   // host checks < 2^31
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t c0 = blockIdx.x * 256 + threadIdx.x; c0 < total; c0 += 2 * stride) {
        uint4 rg[2], rz[2], ry[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint32_t c = c0 + u * stride;
            ok[u] = c < total;
            rg[u] = rz[u] = ry[u] = make_uint4(0, 0, 0, 0);
            if (ok[u]) {
                int64_t off;
                if (DENSE) {
                    off = (int64_t)c * 8;
                } else {
                    const uint32_t p = c >> 2;
                    const uint32_t bf = p / (uint32_t)T, t = p - bf * (uint32_t)T;
                    const uint32_t bb = bf / (uint32_t)F, f = bf - bb * (uint32_t)F;
                    off = bb * zs.sb + f * zs.sf + (int64_t)t * zs.st + ch4 * 8;
                }
                rg[u] = *reinterpret_cast<const uint4*>(gz + off);
                rz[u] = *reinterpret_cast<const uint4*>(z + off);
                ry[u] = *reinterpret_cast<const uint4*>(y2 + (int64_t)c * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float gv[8], zv[8], yv[8];
            unpack8(rg[u], gv);
            unpack8(rz[u], zv);
            unpack8(ry[u], yv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gv[j] = zv[j] > 0.f ? gv[j] : 0.f;
                acc[j] += gv[j];
                acc[8 + j] += gv[j] * yv[j];
            }
            if (ok[u]) *reinterpret_cast<uint4*>(gpre + (int64_t)(c0 + u * stride) * 8) = pack8(gv);
        }
    }
    reduce_ch4<16>(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 4) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][lane][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c4 = threadIdx.x >> 4, k = threadIdx.x & 15;
        const float sm = red[0][c4][k] + red[1][c4][k] + red[2][c4][k] + red[3][c4][k];
        sums[(8 + blockIdx.x) * 64 + (k >> 3) * 32 + c4 * 8 + (k & 7)] = sm;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[kBnCountWord] = __int_as_float((int)gridDim.x);
}

// grid of the reducing kernels: every block ends in one 64-float partial, so few and fat blocks
static int reduce_blocks(int64_t chunks) {
    int64_t b = (chunks + 1023) / 1024;
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

static int stream_blocks(int64_t chunks) {
    int64_t b = (chunks + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

static int check_act(const char* who, int B, int F, int T) {
    SP_CHECK_ARG(B >= 1 && F >= 1 && T >= 1 && B <= 65535, "%s: bad shape B=%d F=%d T=%d", who, B, F, T);
    return SP_OK;
}

}  // namespace sp

using namespace sp;
typedef __hip_bfloat16 bf16;

extern "C" int sp_conv_pack_w_batched(const float* const* w, int L, int bwd_data, void* const* wpack, void* stream) {
    SP_CHECK_ARG(w && wpack && L >= 1 && L <= kConvPackBatch, "sp_conv_pack_w: null pointer or L=%d outside [1,%d]", L, kConvPackBatch);
    SP_CHECK_ARG(bwd_data >= 0 && bwd_data <= 2, "sp_conv_pack_w: bwd_data %d (0 forward, 1 backward-data, 2 both)", bwd_data);
    ConvPackBatch pb;
    for (int l = 0; l < kConvPackBatch; ++l) {
        const int q = l < L ? l : 0;
        pb.w[l] = w[q];
        pb.wpack[l] = (bf16*)wpack[q];
        SP_CHECK_ARG(pb.w[l] && pb.wpack[l], "sp_conv_pack_w: null pointer (weight %d)", q);
    }
    hipLaunchKernelGGL(conv_pack_kernel, dim3(bwd_data == 2 ? 120 : 60, L), dim3(256), 0, (hipStream_t)stream, pb, bwd_data);
    return check_launch("conv_pack_kernel");
}

extern "C" int sp_conv_pack_w(const float* w, int bwd_data, void* wpack, void* stream) {
    return sp_conv_pack_w_batched(&w, 1, bwd_data, &wpack, stream);
}

extern "C" int sp_conv3x5(const void* in, const void* in2, const float* coef, int in_mode, const void* wpack,
                          const float* bias, const void* epi_src, const float* epi_coef, int epi_mode, int B,
                          int F, int T, void* out, float* bn_sums, void* stream) {
    int rc = check_act("sp_conv3x5", B, F, T);
    if (rc) return rc;
    SP_CHECK_ARG(in && wpack && out, "sp_conv3x5: null pointer");
    SP_CHECK_ARG(in_mode == IN_ID || coef, "sp_conv3x5: input transform needs coefficients");
    SP_CHECK_ARG(in_mode != IN_BNBWD || in2, "sp_conv3x5: BN-backward transform needs the second input");
    SP_CHECK_ARG(epi_mode == EPI_NONE || epi_src, "sp_conv3x5: epilogue needs its source tensor");
    SP_CHECK_ARG(epi_mode != EPI_MASK || epi_coef, "sp_conv3x5: mask epilogue needs coefficients");
    const int64_t ntiles = (int64_t)((T + kTP - 1) / kTP) * ((F + kFR - 1) / kFR) * B;
    SP_CHECK_ARG(ntiles < ((int64_t)1 << 30), "sp_conv3x5: too many tiles");
    dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512)), blk(256);   // persistent: 2 workgroups per CU
    hipStream_t s = (hipStream_t)stream;
    SP_CHECK_ARG(!bn_sums || epi_mode != EPI_ADD, "sp_conv3x5: fused statistics exist for epi_mode 0 and 1");
#define SP_CONV_LAUNCH(IM, EP)                                                                          \
    hipLaunchKernelGGL((conv3x5_kernel<IM, EP>), grid, blk, 0, s, (const bf16*)in, (const bf16*)in2, coef, \
                       (const bf16*)wpack, bias, (const bf16*)epi_src, epi_coef, B, F, T, (bf16*)out, bn_sums)
    if (in_mode == IN_ID && epi_mode == EPI_NONE) SP_CONV_LAUNCH(IN_ID, EPI_NONE);
    else if (in_mode == IN_BNRELU && epi_mode == EPI_NONE) SP_CONV_LAUNCH(IN_BNRELU, EPI_NONE);
    else if (in_mode == IN_BNBWD && epi_mode == EPI_MASK) SP_CONV_LAUNCH(IN_BNBWD, EPI_MASK);
    else if (in_mode == IN_BNBWD && epi_mode == EPI_ADD) SP_CONV_LAUNCH(IN_BNBWD, EPI_ADD);
    else return fail(SP_EINVAL, "sp_conv3x5: unsupported mode pair (%d, %d)", in_mode, epi_mode);
#undef SP_CONV_LAUNCH
    return check_launch("conv3x5_kernel");
}

extern "C" int sp_conv3x5_wgrad(const void* a_src, const float* a_coef, int a_mode, const void* g,
                                const void* y, const float* g_coef, int B, int F, int T, float* gwp_ws,
                                float* gw, void* stream) {
    int rc = check_act("sp_conv3x5_wgrad", B, F, T);
    if (rc) return rc;
    SP_CHECK_ARG(a_src && g && y && g_coef && gwp_ws && gw, "sp_conv3x5_wgrad: null pointer");
    SP_CHECK_ARG(a_mode == IN_ID || (a_mode == IN_BNRELU && a_coef), "sp_conv3x5_wgrad: bad activation mode");
    hipStream_t s = (hipStream_t)stream;
    const int fsplit = F >= 8 ? 2 : 1, ntc = (T + 63) / 64;
    const int64_t items = (int64_t)ntc * B * fsplit;
    SP_CHECK_ARG(items < ((int64_t)1 << 30), "sp_conv3x5_wgrad: too many work items");
    // persistent workgroups, one slab of the workspace each: at most SP_CONV_WGRAD_SLABS of them
    const int64_t want = (items + kWG - 1) / kWG;
    const int nwg = (int)(want < SP_CONV_WGRAD_SLABS ? want : SP_CONV_WGRAD_SLABS);
    dim3 grid((unsigned)nwg), blk(320 * kWG);
    if (a_mode == IN_ID)
        hipLaunchKernelGGL((conv3x5_wgrad_kernel<IN_ID>), grid, blk, 0, s, (const bf16*)a_src, a_coef,
                           (const bf16*)g, (const bf16*)y, g_coef, B, F, T, fsplit, ntc, gwp_ws);
    else
        hipLaunchKernelGGL((conv3x5_wgrad_kernel<IN_BNRELU>), grid, blk, 0, s, (const bf16*)a_src, a_coef,
                           (const bf16*)g, (const bf16*)y, g_coef, B, F, T, fsplit, ntc, gwp_ws);
    rc = check_launch("conv3x5_wgrad_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(conv_wgrad_unpack_kernel, dim3(15 * 32 * 32 / 64), dim3(1024), 0, s, gwp_ws, nwg, gw);
    return check_launch("conv_wgrad_unpack_kernel");
}

extern "C" int sp_conv_stem_fwd(const float* spec, const float* w, const float* bias, int B, int Fin, int Tin,
                                void* out, void* stream) {
    SP_CHECK_ARG(B >= 1 && Fin >= 1 && Tin >= 1, "sp_conv_stem_fwd: bad shape");
    SP_CHECK_ARG(spec && w && bias && out, "sp_conv_stem_fwd: null pointer");
    const int F = (Fin - 1) / 2 + 1, T = (Tin - 1) / 2 + 1;
    const int64_t items = (int64_t)B * ((F + kStemRows - 1) / kStemRows) * ((T + 63) / 64);
    SP_CHECK_ARG(items < ((int64_t)1 << 30), "sp_conv_stem_fwd: shape exceeds grid limits");
    const int blocks = items < 768 ? (int)items : 768;   // persistent: three workgroups per CU
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(blocks), dim3(256), 0,
                       (hipStream_t)stream, spec, w, bias, B, Fin, Tin, F, T, (bf16*)out);
    return check_launch("stem_fwd_kernel");
}

extern "C" int sp_conv_stem_bwd(const float* spec, const void* gout, int B, int Fin, int Tin, float* gw,
                                float* gbias, void* stream) {
    SP_CHECK_ARG(B >= 1 && Fin >= 1 && Tin >= 1, "sp_conv_stem_bwd: bad shape");
    SP_CHECK_ARG(spec && gout && gw && gbias, "sp_conv_stem_bwd: null pointer");
    const int F = (Fin - 1) / 2 + 1, T = (Tin - 1) / 2 + 1;
    const int64_t items = (int64_t)B * F * ((T + 63) / 64);
    SP_CHECK_ARG(items < ((int64_t)1 << 30), "sp_conv_stem_bwd: shape exceeds grid limits");
    int blocks = (int)((items + kStemBwdGroups - 1) / kStemBwdGroups);
    if (blocks > 256) blocks = 256;   // one workgroup of twelve waves per CU
    hipLaunchKernelGGL(stem_bwd_kernel, dim3(blocks), dim3(256 * kStemBwdGroups), 0, (hipStream_t)stream, spec,
                       (const bf16*)gout, B, Fin, Tin, F, T, gw, gbias);
    return check_launch("stem_bwd_kernel");
}

extern "C" int sp_bn_stats(const void* y, int64_t npos, float* sums, void* stream) {
    SP_CHECK_ARG(y && sums && npos >= 1, "sp_bn_stats: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nb = reduce_blocks(npos * 4);
    hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(nb), dim3(256), 0, s, (const bf16*)y,
                       (const bf16*)nullptr, npos, sums);
    return check_launch("bn_reduce_kernel<0>");
}

extern "C" int sp_bn_bwd_reduce(const void* g, const void* y, int64_t npos, float* sums, void* stream) {
    SP_CHECK_ARG(g && y && sums && npos >= 1, "sp_bn_bwd_reduce: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int nb = reduce_blocks(npos * 4);
    hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(nb), dim3(256), 0, s, (const bf16*)g,
                       (const bf16*)y, npos, sums);
    return check_launch("bn_reduce_kernel<1>");
}

extern "C" int sp_bn_finalize_fwd(float* sums, int64_t npos, const float* gamma, const float* beta,
                                  float* rmean, float* rvar, float momentum, float eps, int training,
                                  float* coef, float* save, void* stream) {
    SP_CHECK_ARG(gamma && beta && rmean && rvar && coef && save && (sums || !training),
                 "sp_bn_finalize_fwd: null pointer");
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sums, (float)npos,
                       gamma, beta, rmean, rvar, momentum, eps, training, coef, save);
    return check_launch("bn_finalize_fwd_kernel");
}

extern "C" int sp_bn_finalize_bwd(float* sums, const float* save, const float* gamma, int64_t npos,
                                  float* coef3, float* ggamma, float* gbeta, void* stream) {
    SP_CHECK_ARG(sums && save && gamma && coef3 && ggamma && gbeta, "sp_bn_finalize_bwd: null pointer");
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, sums, save, gamma,
                       (float)npos, coef3, ggamma, gbeta);
    return check_launch("bn_finalize_bwd_kernel");
}

extern "C" int sp_block_out(const void* y2, const float* coef, const void* x, int B, int F, int T, void* z,
                            int64_t zsb, int64_t zsf, int64_t zst, void* stream) {
    int rc = check_act("sp_block_out", B, F, T);
    if (rc) return rc;
    SP_CHECK_ARG(y2 && coef && x && z, "sp_block_out: null pointer");
    SP_CHECK_ARG(zsb % 8 == 0 && zsf % 8 == 0 && zst % 8 == 0, "sp_block_out: strides must keep 16-byte chunks aligned");
    Strides zs{zsb, zsf, zst};
    hipLaunchKernelGGL(block_out_kernel, dim3(stream_blocks((int64_t)B * F * T * 4)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)y2, coef, (const bf16*)x, B, F, T, (bf16*)z, zs);
    return check_launch("block_out_kernel");
}

extern "C" int sp_block_bwd_in(const void* gz, const void* z, int64_t zsb, int64_t zsf, int64_t zst,
                               const void* y2, int B, int F, int T, void* gpre, float* sums, void* stream) {
    int rc = check_act("sp_block_bwd_in", B, F, T);
    if (rc) return rc;
    SP_CHECK_ARG(gz && z && y2 && gpre && sums, "sp_block_bwd_in: null pointer");
    SP_CHECK_ARG(zsb % 8 == 0 && zsf % 8 == 0 && zst % 8 == 0, "sp_block_bwd_in: strides must keep 16-byte chunks aligned");
    hipStream_t s = (hipStream_t)stream;
    Strides zs{zsb, zsf, zst};
    SP_CHECK_ARG((int64_t)B * F * T * 4 < ((int64_t)1 << 31), "sp_block_bwd_in: too many positions");
    const bool dense = zst == 32 && zsf == (int64_t)T * 32 && zsb == (int64_t)F * T * 32;
    const int nb = reduce_blocks((int64_t)B * F * T * 4);
    if (dense)
        hipLaunchKernelGGL((block_bwd_in_kernel<true>), dim3(nb), dim3(256), 0, s, (const bf16*)gz, (const bf16*)z, zs,
                           (const bf16*)y2, B, F, T, (bf16*)gpre, sums);
    else
        hipLaunchKernelGGL((block_bwd_in_kernel<false>), dim3(nb), dim3(256), 0, s, (const bf16*)gz, (const bf16*)z, zs,
                           (const bf16*)y2, B, F, T, (bf16*)gpre, sums);
    return check_launch("block_bwd_in_kernel");
}
