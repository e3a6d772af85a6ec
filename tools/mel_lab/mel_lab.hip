// Stand-alone timing harness for the mel frontend kernel (no torch): builds the HTK filterbank of the
// reference configuration (16 kHz, n_fft 400, 80 mels), runs sp_melspec_fwd on B x SECONDS of noise and
// prints the average launch time.  The kernel source is taken from -DKERNEL_SRC="..." so that variants
// (phase cuts, rewrites) are timed in one gpurun call.  Build: see tools/mel_lab/run.sh.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include KERNEL_SRC

namespace sp {
char* err_buf() { static thread_local char buf[512] = {0}; return buf; }
int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(err_buf(), 512, fmt, ap); va_end(ap); return code;
}
}  // namespace sp

static void htk_fbank(std::vector<float>& fb) {
    const int nf = 201, nm = 80; const double sr = 16000.0;
    auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    auto mel2hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
    std::vector<double> pts(nm + 2);
    const double m0 = hz2mel(0.0), m1 = hz2mel(sr / 2);
    for (int i = 0; i < nm + 2; ++i) pts[i] = mel2hz(m0 + (m1 - m0) * i / (nm + 1));
    fb.assign(nf * nm, 0.f);
    for (int f = 0; f < nf; ++f) {
        const double hz = (sr / 2) * f / (nf - 1);
        for (int m = 0; m < nm; ++m) {
            const double up = (hz - pts[m]) / (pts[m + 1] - pts[m]), dn = (pts[m + 2] - hz) / (pts[m + 2] - pts[m + 1]);
            const double v = std::fmax(0.0, std::fmin(up, dn));
            fb[f * nm + m] = (float)v;
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, secs = argc > 2 ? atoi(argv[2]) : 15;
    const bool i16 = argc > 3 && argv[3][0] == 'i';   // third argument "i16": feed int16 PCM (flags bit 2)
    const int N = secs * 16000, T = 1 + N / 160;
    std::vector<float> fb; htk_fbank(fb);
    std::vector<float> wav((size_t)B * N);
    uint32_t s = 12345;
    for (auto& v : wav) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 20001 - 10000) * 1e-5f; }
    std::vector<int64_t> offs(B + 1);
    for (int b = 0; b <= B; ++b) offs[b] = (int64_t)b * N;
    float *d_wav, *d_out; int64_t* d_offs; void* d_plan;
    CK(hipMalloc(&d_wav, wav.size() * 4)); CK(hipMalloc(&d_out, (size_t)B * 80 * T * 4));
    CK(hipMalloc(&d_offs, offs.size() * 8)); CK(hipMalloc(&d_plan, sp_melspec_plan_bytes()));
    if (i16) {
        std::vector<int16_t> pcm(wav.size());
        for (size_t i = 0; i < wav.size(); ++i) pcm[i] = (int16_t)(wav[i] * 32768.f);
        CK(hipMemcpy(d_wav, pcm.data(), pcm.size() * 2, hipMemcpyHostToDevice));
    } else {
        CK(hipMemcpy(d_wav, wav.data(), wav.size() * 4, hipMemcpyHostToDevice));
    }
    const int fl = i16 ? 4 : 0;
    CK(hipMemcpy(d_offs, offs.data(), offs.size() * 8, hipMemcpyHostToDevice));
    if (sp_melspec_plan(fb.data(), d_plan, nullptr)) { printf("plan: %s\n", sp::err_buf()); return 1; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i)
        if (sp_melspec_fwd(d_wav, d_offs, B, T, d_plan, nullptr, fl, d_out, nullptr)) { printf("fwd: %s\n", sp::err_buf()); return 1; }
    CK(hipDeviceSynchronize());
    const int R = 50;
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < R; ++i) sp_melspec_fwd(d_wav, d_offs, B, T, d_plan, nullptr, fl, d_out, nullptr);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> out((size_t)80 * T);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float v : out) cs += v;
    printf("%-32s %s B=%d T=%d  %.2f us/launch  checksum %.6e\n", LABEL, i16 ? "i16" : "f32", B, T, ms * 1000.0 / R, cs);
    return 0;
}
