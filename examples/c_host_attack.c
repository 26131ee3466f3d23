/* The attack-evaluation loop of BASELINE configs 2/3 from a host without Python: plain C against include/advshadow.h.
 *   DDIM sample (diff_model.py:442-474) -> uint8 -> Resize((224, 224)) -> ResNet-50 victim -> argmax -> ASR   (ASR_fast.py:90-126)
 *   apply_shadow closed form on clean images (tools/train_shadow.py:242-266) -> 64 x 64 uint8 -> PSNR / SSIM (PSNR_SSIM_fast.py:21-56)
 * Networks, x_T and the clean images come from an integer generator; tests/test_gpu_handle.py::test_c_host_attack_program rebuilds
 * them in Python and compares predictions and metrics with this package's Python pipeline (attack.attack_shard's stages).
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ examples/c_host_attack.c -Iinclude -I/opt/rocm/include -L<package dir> -ladvshadow_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -o c_host_attack ;  ./c_host_attack [bf16|fp16|fp32] */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "advshadow.h"

static uint32_t g_state = 12345u;
static float next_unit(void) {             /* uniform in [-0.5, 0.5), exact in f32 */
    g_state = g_state * 1664525u + 1013904223u;
    return (float)(g_state >> 8) * (1.0f / 16777216.0f) - 0.5f;
}
static int ends_with(const char* s, const char* suf) {
    const size_t a = strlen(s), b = strlen(suf);
    return a >= b && !strcmp(s + a - b, suf);
}
#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, advs_last_error()); return 1; } } while (0)
#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)
#define B 4
#define S 64
#define STEPS 4
#define CLASSES 37

int main(int argc, char** argv) {
    const char* dts = argc > 1 ? argv[1] : "bf16";
    const int dt = !strcmp(dts, "fp32") ? ADVS_F32 : (!strcmp(dts, "fp16") ? ADVS_F16 : ADVS_BF16);
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));

    /* ---- the eps-predictor */
    advs_unet_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.in_channels = 3; cfg.model_channels = 64; cfg.out_channels = 3; cfg.num_res_blocks = 1;
    cfg.n_attention_resolutions = 1; cfg.attention_resolutions[0] = 2;
    cfg.n_channel_mult = 2; cfg.channel_mult[0] = 1; cfg.channel_mult[1] = 2;
    cfg.num_heads = 4; cfg.dtype = dt;
    advs_unet* net = NULL;
    CHECK(advs_unet_create(&cfg, &net));
    for (int i = 0; i < advs_unet_param_count(net); ++i) {
        char name[256];
        long long n = 0;
        CHECK(advs_unet_param_name(net, i, name, sizeof(name), &n));
        float* v = (float*)malloc((size_t)n * sizeof(float));
        const int is_bias = ends_with(name, ".bias");
        const int is_norm_w = !is_bias && (strstr(name, ".conv1.0.") || strstr(name, ".conv2.0.") || strstr(name, ".norm.") || !strncmp(name, "out.0.", 6));
        for (long long k = 0; k < n; ++k) {
            const float u = next_unit();
            v[k] = is_norm_w ? 1.0f + 0.25f * u : (is_bias ? 0.125f * u : 0.25f * u);
        }
        CHECK(advs_unet_set_param(net, name, v, n));
        free(v);
    }
    /* ---- the victim */
    advs_resnet50* vic = NULL;
    CHECK(advs_resnet50_create(CLASSES, dt, &vic));
    for (int i = 0; i < advs_resnet50_param_count(vic); ++i) {
        char name[256];
        long long n = 0;
        CHECK(advs_resnet50_param_name(vic, i, name, sizeof(name), &n));
        float* v = (float*)malloc((size_t)n * sizeof(float));
        const int is_bn = strstr(name, "bn") != NULL || strstr(name, "downsample.1.") != NULL;
        for (long long k = 0; k < n; ++k) {
            const float u = next_unit();
            if (ends_with(name, "running_var")) v[k] = 1.0f + 0.5f * u;
            else if (ends_with(name, "running_mean")) v[k] = 0.25f * u;
            else if (is_bn && ends_with(name, ".weight")) v[k] = 1.0f + 0.25f * u;
            else if (ends_with(name, ".bias")) v[k] = 0.125f * u;
            else if (strstr(name, "downsample.0.")) v[k] = 0.25f * u;
            else if (!strncmp(name, "fc.", 3)) v[k] = 0.25f * u;
            else v[k] = 0.125f * u;                         /* conv kernels */
        }
        CHECK(advs_resnet50_set_param(vic, name, v, n));
        free(v);
    }
    CHECK(advs_unet_plan(net, B, S, 1, st));
    CHECK(advs_resnet50_plan(vic, B, 224, S, st));

    /* ---- sample */
    int n = 0;
    CHECK(advs_ddim_tables(1, 1000, STEPS, 0, 0.0f, NULL, NULL, &n));
    float* coef = (float*)malloc((size_t)n * 3 * sizeof(float));
    int64_t* tseq = (int64_t*)malloc((size_t)n * sizeof(int64_t));
    CHECK(advs_ddim_tables(1, 1000, STEPS, 0, 0.0f, coef, tseq, &n));
    const size_t cnt = (size_t)B * 3 * S * S;
    float* xh = (float*)malloc(cnt * sizeof(float));
    for (size_t k = 0; k < cnt; ++k) xh[k] = 4.0f * next_unit();
    float *xd = NULL, *clean = NULL, *shadowed = NULL, *mask = NULL, *cen = NULL, *rad = NULL, *c64 = NULL, *s64 = NULL;
    uint8_t *gen = NULL, *u8a = NULL, *u8b = NULL;
    int* pred = NULL;
    double* sp = NULL;
    HIPCHECK(hipMalloc((void**)&xd, cnt * sizeof(float)));
    HIPCHECK(hipMemcpy(xd, xh, cnt * sizeof(float), hipMemcpyHostToDevice));
    CHECK(advs_ddim_run(net, xd, coef, tseq, n, 1));
    /* ---- uint8 cast (clamped) and the victim */
    HIPCHECK(hipMalloc((void**)&gen, cnt));
    HIPCHECK(hipMalloc((void**)&pred, B * sizeof(int)));
    CHECK(advs_to_uint8(xd, gen, cnt, 1, st));
    CHECK(advs_resnet50_eval_u8(vic, gen, pred));
    /* ---- shadow composite on clean images, 64 x 64 PSNR / SSIM */
    for (size_t k = 0; k < cnt; ++k) xh[k] = next_unit() + 0.5f;             /* clean images in [0, 1) */
    float hm[B * S * S], hc[B * 2], hr[B];
    for (int k = 0; k < B * S * S; ++k) hm[k] = 1.0f;
    for (int b = 0; b < B; ++b) { hc[2 * b] = 20.0f + 6.0f * b; hc[2 * b + 1] = 40.0f - 5.0f * b; hr[b] = 10.0f + 2.0f * b; }
    const float ht[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f};              /* cv2.getGaussianKernel(5, 0); a HOST array for advs_apply_shadow */
    HIPCHECK(hipMalloc((void**)&clean, cnt * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&shadowed, cnt * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&mask, sizeof(hm)));
    HIPCHECK(hipMalloc((void**)&cen, sizeof(hc)));
    HIPCHECK(hipMalloc((void**)&rad, sizeof(hr)));
    HIPCHECK(hipMalloc((void**)&c64, cnt * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&s64, cnt * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&u8a, cnt));
    HIPCHECK(hipMalloc((void**)&u8b, cnt));
    HIPCHECK(hipMalloc((void**)&sp, B * 2 * sizeof(double)));
    HIPCHECK(hipMemcpyAsync(clean, xh, cnt * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(mask, hm, sizeof(hm), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(cen, hc, sizeof(hc), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(rad, hr, sizeof(hr), hipMemcpyHostToDevice, st));
    CHECK(advs_apply_shadow(clean, mask, cen, rad, shadowed, B, 3, S, S, 1, 0.43f, ht, 5, st));
    /* what PSNR_SSIM_fast.load_image does to a saved image: uint8 -> Resize((64, 64)) (the identity here) -> ToTensor */
    CHECK(advs_unit_to_uint8(clean, u8a, cnt, st));
    CHECK(advs_u8_nchw_to_hwc(u8a, u8b, B, 3, S, S, st));
    CHECK(advs_u8hwc_to_f32nchw(u8b, c64, B, S, S, 3, NULL, NULL, st));
    CHECK(advs_unit_to_uint8(shadowed, u8a, cnt, st));
    CHECK(advs_u8_nchw_to_hwc(u8a, u8b, B, 3, S, S, st));
    CHECK(advs_u8hwc_to_f32nchw(u8b, s64, B, S, S, 3, NULL, NULL, st));
    CHECK(advs_psnr_ssim(c64, s64, sp, B, 3, S, S, 7, st));
    HIPCHECK(hipStreamSynchronize(st));
    int hp[B];
    double hs[B * 2];
    HIPCHECK(hipMemcpy(hp, pred, sizeof(hp), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(hs, sp, sizeof(hs), hipMemcpyDeviceToHost));
    int wrong = 0;
    for (int b = 0; b < B; ++b) wrong += hp[b] != (b * 7) % CLASSES;          /* labels of the synthetic set */
    printf("dtype %s asr %.6f\n", dts, (double)wrong / B);
    for (int b = 0; b < B; ++b) printf("image %d pred %d ssim %.17g psnr %.17g\n", b, hp[b], hs[2 * b], hs[2 * b + 1]);
    advs_unet_destroy(net);
    advs_resnet50_destroy(vic);
    return 0;
}
