/* A host without Python: plain C against include/advshadow.h and the HIP runtime's C API.
 * Builds a small eps-predictor (diff_model.UNetModel's constructor arguments), fills its state_dict from an integer generator,
 * and runs the DDIM reverse loop (diff_model.py:442-474) on the GPU through libadvshadow_hip.so alone.
 *   hipcc -x c examples/c_host_ddim.c -Iinclude -L<package dir> -ladvshadow_hip -Wl,-rpath,<package dir> -o c_host_ddim
 *   ./c_host_ddim [bf16|fp16|fp32] [steps]      -> one line: dtype, steps, n, sum, sum of |x|, the first four values as hex
 * tests/test_gpu_handle.py::test_c_host_program rebuilds the same weights in Python and compares the sample bit for bit. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "advshadow.h"

static uint32_t g_state = 12345u;
static float next_unit(void) {             /* uniform in [-0.5, 0.5), exact in f32: 24 bits of a 32-bit LCG */
    g_state = g_state * 1664525u + 1013904223u;
    return (float)(g_state >> 8) * (1.0f / 16777216.0f) - 0.5f;
}
#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, advs_last_error()); return 1; } } while (0)
#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const char* dts = argc > 1 ? argv[1] : "bf16";
    const int steps = argc > 2 ? atoi(argv[2]) : 5;
    const int B = 2, S = 32;
    advs_unet_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.in_channels = 3; cfg.model_channels = 64; cfg.out_channels = 3; cfg.num_res_blocks = 1;
    cfg.n_attention_resolutions = 1; cfg.attention_resolutions[0] = 2;
    cfg.n_channel_mult = 2; cfg.channel_mult[0] = 1; cfg.channel_mult[1] = 2;
    cfg.num_heads = 4;
    cfg.dtype = !strcmp(dts, "fp32") ? ADVS_F32 : (!strcmp(dts, "fp16") ? ADVS_F16 : ADVS_BF16);
    advs_unet* net = NULL;
    CHECK(advs_unet_create(&cfg, &net));
    const int np = advs_unet_param_count(net);
    for (int i = 0; i < np; ++i) {
        char name[256];
        long long n = 0;
        CHECK(advs_unet_param_name(net, i, name, sizeof(name), &n));
        float* v = (float*)malloc((size_t)n * sizeof(float));
        const size_t len = strlen(name);
        const int is_bias = len > 5 && !strcmp(name + len - 5, ".bias");
        /* GroupNorm weights (1-D ".weight" tensors) around 1, biases small, matrices and kernels ~ +-0.1 */
        const int is_norm_w = !is_bias && (strstr(name, ".conv1.0.") || strstr(name, ".conv2.0.") || strstr(name, ".norm.") || !strncmp(name, "out.0.", 6));
        for (long long k = 0; k < n; ++k) {
            const float u = next_unit();
            v[k] = is_norm_w ? 1.0f + 0.25f * u : (is_bias ? 0.125f * u : 0.25f * u);
        }
        CHECK(advs_unet_set_param(net, name, v, n));
        free(v);
    }
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));
    CHECK(advs_unet_plan(net, B, S, 1, st));
    int n = 0;
    CHECK(advs_ddim_tables(1, 1000, steps, 0, 0.0f, NULL, NULL, &n));
    float* coef = (float*)malloc((size_t)n * 3 * sizeof(float));
    int64_t* tseq = (int64_t*)malloc((size_t)n * sizeof(int64_t));
    CHECK(advs_ddim_tables(1, 1000, steps, 0, 0.0f, coef, tseq, &n));
    const size_t cnt = (size_t)B * 3 * S * S;
    float* xh = (float*)malloc(cnt * sizeof(float));
    for (size_t k = 0; k < cnt; ++k) xh[k] = 4.0f * next_unit();           /* x_T */
    float* xd = NULL;
    HIPCHECK(hipMalloc((void**)&xd, cnt * sizeof(float)));
    HIPCHECK(hipMemcpy(xd, xh, cnt * sizeof(float), hipMemcpyHostToDevice));
    CHECK(advs_ddim_run(net, xd, coef, tseq, n, 1));
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipMemcpy(xh, xd, cnt * sizeof(float), hipMemcpyDeviceToHost));
    double s = 0.0, sa = 0.0;
    for (size_t k = 0; k < cnt; ++k) { s += xh[k]; sa += xh[k] < 0 ? -xh[k] : xh[k]; }
    uint32_t h[4];
    memcpy(h, xh, sizeof(h));
    printf("%s steps %d n %zu sum %.9e abs %.9e first %08x %08x %08x %08x\n", dts, n, cnt, s, sa, h[0], h[1], h[2], h[3]);
    if (argc > 3) {                         /* raw f32 dump for a bit-for-bit comparison */
        FILE* f = fopen(argv[3], "wb");
        if (!f || fwrite(xh, sizeof(float), cnt, f) != cnt) { fprintf(stderr, "cannot write %s\n", argv[3]); return 1; }
        fclose(f);
    }
    advs_unet_destroy(net);
    (void)hipFree(xd);
    (void)hipStreamDestroy(st);
    free(coef); free(tseq); free(xh);
    return 0;
}
