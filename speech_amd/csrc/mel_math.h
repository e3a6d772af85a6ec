// FFT building blocks of the mel frontend, written once for device and host: the host build
// (tests/host/mel_emulate.cpp, g++) replays the kernel's phases lane by lane so that the index
// arithmetic and twiddles are checked on the CPU before a GPU ever runs them.
//
// 400-point real FFT of a windowed frame x[0..400):
//   z[n] = x[2n] + i x[2n+1]            (N = 200 complex points)
//   200 = 8 * 25, n = 25 n1 + n2, k = k1 + 8 k2:
//     phase 1  A[n2][k1] = sum_n1 z[25 n1 + n2] W8^(n1 k1)            (8-point DFT per n2)
//              Y[k1][n2] = A[n2][k1] * W200^(n2 k1)
//     phase 2  Z[k1 + 8 k2] = sum_n2 Y[k1][n2] W25^(n2 k2)            (25-point DFT per k1, 5 x 5)
//     phase 3  E = (Z[k] + conj Z[200-k]) / 2, O = -i (Z[k] - conj Z[200-k]) / 2
//              X[k] = E + W400^k O,  X[200-k] = conj(E - W400^k O),   k = 0..100
#pragma once

#if defined(__HIPCC__)
#define SP_HD __host__ __device__ __forceinline__
#else
#define SP_HD inline
#endif

namespace sp {
namespace melmath {

struct cf {
    float re, im;
};

SP_HD cf mk(float re, float im) { cf r; r.re = re; r.im = im; return r; }
SP_HD cf add(cf a, cf b) { return mk(a.re + b.re, a.im + b.im); }
SP_HD cf sub(cf a, cf b) { return mk(a.re - b.re, a.im - b.im); }
SP_HD cf mul(cf a, cf b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
SP_HD cf mul_negi(cf a) { return mk(a.im, -a.re); }  // a * (-i)
SP_HD cf scale(cf a, float s) { return mk(a.re * s, a.im * s); }

constexpr float kInvSqrt2 = 0.70710678118654752440f;
// cos/sin of 2*pi/5 and 4*pi/5
constexpr float kC1 = 0.30901699437494742410f;
constexpr float kC2 = -0.80901699437494742410f;
constexpr float kS1 = 0.95105651629515357212f;
constexpr float kS2 = 0.58778525229247312917f;

// Forward 8-point DFT in place: z[k] <- sum_n z[n] exp(-2 pi i n k / 8)
SP_HD void dft8(cf (&z)[8]) {
    const cf a0 = add(z[0], z[4]), a4 = sub(z[0], z[4]);
    const cf a1 = add(z[1], z[5]), a5 = sub(z[1], z[5]);
    const cf a2 = add(z[2], z[6]), a6 = sub(z[2], z[6]);
    const cf a3 = add(z[3], z[7]), a7 = sub(z[3], z[7]);
    // even outputs
    const cf b0 = add(a0, a2), b2 = sub(a0, a2);
    const cf b1 = add(a1, a3), b3 = mul_negi(sub(a1, a3));
    // odd outputs: twiddle a5 by W8, a6 by -i, a7 by W8^3
    const cf c0 = a4;
    const cf c1 = mk((a5.re + a5.im) * kInvSqrt2, (a5.im - a5.re) * kInvSqrt2);   // a5 * (1 - i)/sqrt2
    const cf c2 = mul_negi(a6);
    const cf c3 = mk((a7.im - a7.re) * kInvSqrt2, -(a7.re + a7.im) * kInvSqrt2);  // a7 * (-1 - i)/sqrt2
    const cf d0 = add(c0, c2), d2 = sub(c0, c2);
    const cf d1 = add(c1, c3), d3 = mul_negi(sub(c1, c3));
    z[0] = add(b0, b1); z[4] = sub(b0, b1); z[2] = add(b2, b3); z[6] = sub(b2, b3);
    z[1] = add(d0, d1); z[5] = sub(d0, d1); z[3] = add(d2, d3); z[7] = sub(d2, d3);
}

// Forward 5-point DFT: (x0..x4) -> (y0..y4)
SP_HD void dft5(cf x0, cf x1, cf x2, cf x3, cf x4, cf& y0, cf& y1, cf& y2, cf& y3, cf& y4) {
    const cf t1 = add(x1, x4), t2 = add(x2, x3), t3 = sub(x1, x4), t4 = sub(x2, x3);
    y0 = add(x0, add(t1, t2));
    const cf a1 = mk(x0.re + kC1 * t1.re + kC2 * t2.re, x0.im + kC1 * t1.im + kC2 * t2.im);
    const cf a2 = mk(x0.re + kC2 * t1.re + kC1 * t2.re, x0.im + kC2 * t1.im + kC1 * t2.im);
    const cf b1 = mk(kS1 * t3.re + kS2 * t4.re, kS1 * t3.im + kS2 * t4.im);
    const cf b2 = mk(kS2 * t3.re - kS1 * t4.re, kS2 * t3.im - kS1 * t4.im);
    // y1 = a1 - i b1, y4 = a1 + i b1, y2 = a2 - i b2, y3 = a2 + i b2
    y1 = mk(a1.re + b1.im, a1.im - b1.re);
    y4 = mk(a1.re - b1.im, a1.im + b1.re);
    y2 = mk(a2.re + b2.im, a2.im - b2.re);
    y3 = mk(a2.re - b2.im, a2.im + b2.re);
}

// W25^(b*c) for b, c in 1..4 (row b-1, col c-1), as (cos, -sin) of 2 pi b c / 25
#define SP_TW25_RE \
    {{0.96858316112863107605f, 0.87630668004386358394f, 0.72896862742141155245f, 0.53582679497899654564f}, \
     {0.87630668004386358394f, 0.53582679497899654564f, 0.06279051952931352654f, -0.42577929156507271502f}, \
     {0.72896862742141155245f, 0.06279051952931352654f, -0.63742398974868974548f, -0.99211470131447776488f}, \
     {0.53582679497899654564f, -0.42577929156507271502f, -0.99211470131447776488f, -0.63742398974868952344f}}
#define SP_TW25_IM \
    {{-0.24868988716485479484f, -0.48175367410171532345f, -0.68454710592868861507f, -0.84432792550201507531f}, \
     {-0.48175367410171532345f, -0.84432792550201507531f, -0.99802672842827155897f, -0.90482705246601946580f}, \
     {-0.68454710592868861507f, -0.99802672842827155897f, -0.77051324277578925326f, -0.12533323356430453588f}, \
     {-0.84432792550201507531f, -0.90482705246601946580f, -0.12533323356430453588f, 0.77051324277578936428f}}

// Forward 25-point DFT in place: y[k2] <- sum_n2 y[n2] W25^(n2 k2).
// n2 = 5 a + b, k2 = c + 5 d:  out[c + 5 d] = sum_b W5^(b d) * ( W25^(b c) * sum_a y[5 a + b] W5^(a c) )
SP_HD void dft25(cf (&y)[25]) {
    constexpr float twre[4][4] = SP_TW25_RE;
    constexpr float twim[4][4] = SP_TW25_IM;
    cf g[5][5];  // g[b][c]
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 0; b < 5; ++b) {
        dft5(y[b], y[5 + b], y[10 + b], y[15 + b], y[20 + b], g[b][0], g[b][1], g[b][2], g[b][3], g[b][4]);
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int b = 1; b < 5; ++b) {
#if defined(__HIPCC__)
#pragma unroll
#endif
        for (int c = 1; c < 5; ++c) g[b][c] = mul(g[b][c], mk(twre[b - 1][c - 1], twim[b - 1][c - 1]));
    }
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int c = 0; c < 5; ++c) {
        dft5(g[0][c], g[1][c], g[2][c], g[3][c], g[4][c], y[c], y[c + 5], y[c + 10], y[c + 15], y[c + 20]);
    }
}

// reflect-pad index map for a centred frame: sample index i in [-200, N+200) -> [0, N)
SP_HD int reflect(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// phase 3 for one k in 0..100: from Zk = Z[k] and Zm = Z[200-k] (Z[200] := Z[0]) and w = W400^k,
// the magnitudes |X[k]| and |X[200-k]|
SP_HD void real_post(cf Zk, cf Zm, cf w, float& mag_k, float& mag_m) {
    const cf zc = mk(Zm.re, -Zm.im);
    const cf E = scale(add(Zk, zc), 0.5f);
    const cf O = scale(mul_negi(sub(Zk, zc)), 0.5f);
    const cf wo = mul(w, O);
    const cf xa = add(E, wo), xb = sub(E, wo);
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    // v_sqrt_f32 (1 ulp); the IEEE expansion is ten instructions per magnitude
    mag_k = __builtin_amdgcn_sqrtf(xa.re * xa.re + xa.im * xa.im);
    mag_m = __builtin_amdgcn_sqrtf(xb.re * xb.re + xb.im * xb.im);
#else
    mag_k = __builtin_sqrtf(xa.re * xa.re + xa.im * xa.im);
    mag_m = __builtin_sqrtf(xb.re * xb.re + xb.im * xb.im);
#endif
}

}  // namespace melmath
}  // namespace sp
