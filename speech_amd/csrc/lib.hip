// libspeech_hip.so: error reporting, version, and the host-side pieces of the C ABI.
#include <vector>

#include "common.h"

namespace sp {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace sp

extern "C" const char* sp_last_error(void) { return sp::err_buf(); }
extern "C" int sp_version(void) { return 100; }

// Unit-cost Levenshtein distance (decoder.py:17, :25 call Levenshtein.distance).  Host code:
// the strings are a few hundred symbols and live on the host after decode.
extern "C" int sp_edit_distance(const int32_t* a, int na, const int32_t* b, int nb) {
    if (na < 0 || nb < 0) return SP_EINVAL;
    if (na < nb) {
        const int32_t* tp = a; a = b; b = tp;
        int tn = na; na = nb; nb = tn;
    }
    std::vector<int> prev(nb + 1), cur(nb + 1);
    for (int j = 0; j <= nb; ++j) prev[j] = j;
    for (int i = 1; i <= na; ++i) {
        cur[0] = i;
        const int32_t ca = a[i - 1];
        for (int j = 1; j <= nb; ++j) {
            const int sub = prev[j - 1] + (ca != b[j - 1]);
            const int del = prev[j] + 1, ins = cur[j - 1] + 1;
            int m = sub < del ? sub : del;
            cur[j] = m < ins ? m : ins;
        }
        prev.swap(cur);
    }
    return prev[nb];
}
