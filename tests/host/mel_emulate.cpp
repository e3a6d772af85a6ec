// Host replay of the mel kernel's phases (speech_amd/csrc/mel.hip) using the SAME math header
// (mel_math.h): every lane-task of phases 1-3 is executed sequentially on plain arrays, so the
// 8 x 25 index split, the twiddle tables and the real-FFT post-processing are checked on the CPU.
// Usage: mel_emulate <wav.f32> <fbank.f32 [201*80]> <out.f32 [80*T]>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../speech_amd/csrc/mel_math.h"

using namespace sp::melmath;

int main(int argc, char** argv) {
    if (argc != 4) return 2;
    auto slurp = [](const char* p) {
        std::vector<float> v;
        FILE* f = fopen(p, "rb");
        if (!f) exit(3);
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        v.resize(n / 4);
        if (fread(v.data(), 4, v.size(), f) != v.size()) exit(4);
        fclose(f);
        return v;
    };
    std::vector<float> wav = slurp(argv[1]), fb = slurp(argv[2]);
    const int N = (int)wav.size();
    const int T = 1 + N / 160;
    const double two_pi = 6.283185307179586476925286766559;
    float win[400];
    for (int n = 0; n < 400; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(two_pi * n / 400));
    cf tw200[25][8], tw400[101];
    for (int n2 = 0; n2 < 25; ++n2)
        for (int k1 = 0; k1 < 8; ++k1) {
            double a = two_pi * (n2 * k1) / 200.0;
            tw200[n2][k1] = mk((float)std::cos(a), (float)-std::sin(a));
        }
    for (int k = 0; k <= 100; ++k) {
        double a = two_pi * k / 400.0;
        tw400[k] = mk((float)std::cos(a), (float)-std::sin(a));
    }
    std::vector<float> out((size_t)80 * T);
    std::vector<cf> yz(200), tmp(200);
    std::vector<float> mag(201);
    for (int t = 0; t < T; ++t) {
        // phase 1: tasks n2 = 0..24
        for (int n2 = 0; n2 < 25; ++n2) {
            cf z[8];
            const int base = t * 160 - 200 + 2 * n2;
            for (int n1 = 0; n1 < 8; ++n1) {
                const int m2 = 50 * n1 + 2 * n2;
                const int i0 = reflect(base + 50 * n1, N), i1 = reflect(base + 50 * n1 + 1, N);
                z[n1] = mk(wav[i0] * win[m2], wav[i1] * win[m2 + 1]);
            }
            dft8(z);
            yz[n2] = z[0];
            for (int k1 = 1; k1 < 8; ++k1) yz[k1 * 25 + n2] = mul(z[k1], tw200[n2][k1]);
        }
        // phase 2: tasks k1 = 0..7 (all reads before all writes, as in the wave)
        for (int k1 = 0; k1 < 8; ++k1) {
            cf y[25];
            for (int n2 = 0; n2 < 25; ++n2) y[n2] = yz[k1 * 25 + n2];
            dft25(y);
            for (int k2 = 0; k2 < 25; ++k2) tmp[k1 + 8 * k2] = y[k2];
        }
        yz = tmp;
        // phase 3
        for (int k = 0; k <= 100; ++k) {
            const int km = k == 0 ? 0 : 200 - k;
            float a, b;
            real_post(yz[k], yz[km], tw400[k], a, b);
            mag[k] = a;
            mag[200 - k] = b;
        }
        // phase 4: dense projection in bin order (the MFMA chain is a k-ordered fmaf chain)
        for (int m = 0; m < 80; ++m) {
            float acc = 0.f;
            for (int f = 0; f < 201; ++f) acc = std::fmaf(fb[f * 80 + m], mag[f], acc);
            out[(size_t)m * T + t] = acc;
        }
    }
    FILE* f = fopen(argv[3], "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    return 0;
}
