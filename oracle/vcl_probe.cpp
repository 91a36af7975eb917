// vcl_probe.cpp -- prints results of the reference's OWN Vec4d primitives (compiled from
// /root/reference/vectorclass, -mavx, no -mfma: the reference's default AVX build flags,
// CMakeLists.txt:360-362) for fixed inputs.  tests/make_vcl_golden.py stores the output as
// tests/golden/vcl_probe.json; tests/test_vcl_probe.py checks that the oracle's scalar
// emulation (dot4 association, unfused mul_add, libm exp/log vs VCL exp/log) agrees.
#include <stdio.h>
#include <stdlib.h>
#include "vectorclass.h"
#include "vectormath_exp.h"

static unsigned long long s = 88172645463325252ULL;
static double rnd() {  // xorshift64*, uniform in (-1, 1)
    s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
    unsigned long long r = s * 2685821657736338717ULL;
    return ((double)(r >> 11) / 9007199254740992.0) * 2.0 - 1.0;
}

int main() {
    printf("{\n");
    // 1) mul_add chains + horizontal_add: an n-term dot product exactly as the kernels do it
    printf("\"dots\": [\n");
    int sizes[4] = {4, 20, 64, 16};
    for (int t = 0; t < 40; t++) {
        int n = sizes[t % 4];
        double a[64], b[64];
        for (int i = 0; i < n; i++) { a[i] = rnd() * 1e-3; b[i] = rnd(); }
        Vec4d acc = Vec4d().load(a) * Vec4d().load(b);
        for (int i = 4; i < n; i += 4) acc = mul_add(Vec4d().load(a + i), Vec4d().load(b + i), acc);
        double r = horizontal_add(acc);
        printf("{\"n\": %d, \"a\": [", n);
        for (int i = 0; i < n; i++) printf("%s%.17g", i ? "," : "", a[i]);
        printf("], \"b\": [");
        for (int i = 0; i < n; i++) printf("%s%.17g", i ? "," : "", b[i]);
        printf("], \"r\": %.17g}%s\n", r, t == 39 ? "" : ",");
    }
    printf("],\n\"exp\": [\n");
    // 2) exp on the range the kernels use: eval*rate*len in [-60, 0]
    for (int t = 0; t < 64; t++) {
        double x[4];
        for (int k = 0; k < 4; k++) x[k] = -(rnd() + 1.0) * 0.5 * (t < 32 ? 5.0 : 60.0);
        Vec4d e = exp(Vec4d().load(x));
        for (int k = 0; k < 4; k++)
            printf("[%.17g, %.17g]%s\n", x[k], e[k], (t == 63 && k == 3) ? "" : ",");
    }
    printf("],\n\"log\": [\n");
    // 3) log on pattern-likelihood magnitudes 1e-300 .. 1
    for (int t = 0; t < 64; t++) {
        double x[4];
        for (int k = 0; k < 4; k++) x[k] = exp((rnd() - 1.0) * 0.5 * 690.0);
        Vec4d l = log(Vec4d().load(x));
        for (int k = 0; k < 4; k++)
            printf("[%.17g, %.17g]%s\n", x[k], l[k], (t == 63 && k == 3) ? "" : ",");
    }
    printf("],\n\"dots8f\": [\n");
    // 4) the UFBoot / RELL dot product: dotProductSIMD<float, Vec8f, 8> (phylokernel.h:55-61) on
    //    pattern-lnL-like x (negative) and bootstrap-count-like y (small integers), zero padded to x8
    int sizes8[5] = {8, 24, 64, 360, 1000};
    for (int t = 0; t < 10; t++) {
        int n = sizes8[t % 5];
        static float x[1008] __attribute__((aligned(32))), y[1008] __attribute__((aligned(32)));
        for (int i = 0; i < n; i++) { x[i] = (float)(-(rnd() + 1.0) * 25.0 - 0.5); y[i] = (float)(int)((rnd() + 1.0) * 2.5); }
        Vec8f res = Vec8f().load_a(x) * Vec8f().load_a(y);
        for (int i = 8; i < n; i += 8) res = mul_add(Vec8f().load_a(&x[i]), Vec8f().load_a(&y[i]), res);
        float r = horizontal_add(res);
        printf("{\"n\": %d, \"x\": [", n);
        for (int i = 0; i < n; i++) printf("%s%.9g", i ? "," : "", x[i]);
        printf("], \"y\": [");
        for (int i = 0; i < n; i++) printf("%s%.9g", i ? "," : "", y[i]);
        printf("], \"r\": %.9g}%s\n", r, t == 9 ? "" : ",");
    }
    printf("]\n}\n");
    return 0;
}
