// External contours of a binary mask on the device: what add_shadow.py:40-47 and shadow_for_attack.py:30-35 ask of
// cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) + cv2.contourArea + cv2.boundingRect, for a batch of masks.
//
// OpenCV (Suzuki-Abe border following, absent from this image: semantics restated, parity unpinned, see oracle/contours.py)
// returns one OUTER border per 8-connected foreground component that is not nested inside a hole of another component.
// Equivalent formulation used here, without tracing:
//   1. label the BACKGROUND with 4-connectivity; background that reaches the image frame gets label 0;
//   2. X = every pixel that is not frame-connected background = the external components with their holes (and anything nested
//      in them) filled.  The 8-connected components of X are exactly the external contours' interiors;
//   3. per component: first pixel in raster order (= where Suzuki's raster scan meets the border, i.e. discovery order),
//      bounding box, pixel count, and contourArea.  The outer border polygon runs through the centres of the border pixels, so
//      its shoelace area is a sum over the 2x2 cells of pixel centres: a cell with all four corners in the component adds 1,
//      with exactly three corners 1/2, anything else 0 (two diagonal corners are a one-pixel bridge the border walks twice).
// Labelling is label-equivalence (Hawick et al.): min over neighbours -> atomicMin on the label's root -> path flattening,
// repeated to a fixed point; a component's final label is its first raster pixel + 1.  One workgroup (1024 threads) per image,
// labels in global memory (L2-resident: 8 bytes per pixel), a few passes per image; bytes: ~40 B per pixel.  Integer work,
// exact.
#include "common.h"

#define CT_THREADS 1024

// one label-equivalence run over `lab` (-1 = not a member, 0 = frame, p + 1 otherwise); conn8: 8- or 4-neighbourhood
__device__ __forceinline__ void label_fixed_point(int* lab, int H, int W, bool conn8, int* s_changed) {
    const int HW = H * W, tid = threadIdx.x;
    for (;;) {
        __syncthreads();
        if (tid == 0) *s_changed = 0;
        __syncthreads();
        for (int p = tid; p < HW; p += CT_THREADS) {
            const int l = lab[p];
            if (l <= 0) continue;
            const int y = p / W, x = p - y * W;
            int mn = l;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if ((dy == 0 && dx == 0) || (!conn8 && dy != 0 && dx != 0)) continue;
                    const int yy = y + dy, xx = x + dx;
                    if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                        const int q = lab[yy * W + xx];
                        if (q >= 0 && q < mn) mn = q;
                    }
                }
            if (mn < l) { atomicMin(&lab[l - 1], mn); *s_changed = 1; }     // lower the label's root pixel
        }
        __syncthreads();
        for (int p = tid; p < HW; p += CT_THREADS) {                        // flatten: follow the references to the root
            int r = lab[p];
            if (r <= 0) continue;
            int guard = 0;
            while (r > 0 && lab[r - 1] != r && guard++ < HW) r = lab[r - 1];
            lab[p] = r;
        }
        __syncthreads();
        if (*s_changed == 0) break;                                         // uniform: every thread reads the same word
    }
}

// out[img][k][8] = {first pixel (raster index), min x, min y, max x, max y, pixel count, twice the contour area, 0}
__global__ void __launch_bounds__(CT_THREADS)
mask_contours_kernel(const uint8_t* __restrict__ mask, int H, int W, int* __restrict__ work, int* __restrict__ out,
                     int* __restrict__ count, int maxc) {
    __shared__ int s_changed, s_count;
    const int img = blockIdx.x, tid = threadIdx.x, HW = H * W;
    const uint8_t* m = mask + (size_t)img * HW;
    int* lb = work + (size_t)img * 2 * HW;         // background labels, later: root pixel -> compact component id
    int* lf = lb + HW;                             // labels of X
    int* o = out + (size_t)img * maxc * 8;
    // ---- 1. background, 4-connectivity, frame = 0
    for (int p = tid; p < HW; p += CT_THREADS) {
        const int y = p / W, x = p - y * W;
        const bool edge = y == 0 || x == 0 || y == H - 1 || x == W - 1;
        lb[p] = m[p] ? -1 : (edge ? 0 : p + 1);
    }
    label_fixed_point(lb, H, W, false, &s_changed);
    // ---- 2. X = not frame-connected background, 8-connectivity
    for (int p = tid; p < HW; p += CT_THREADS) lf[p] = (m[p] == 0 && lb[p] == 0) ? -1 : p + 1;
    label_fixed_point(lf, H, W, true, &s_changed);
    // ---- 3. compact ids for the roots, then per-component statistics
    if (tid == 0) s_count = 0;
    __syncthreads();
    for (int p = tid; p < HW; p += CT_THREADS) {
        lb[p] = -1;
        if (lf[p] == p + 1) {
            const int id = atomicAdd(&s_count, 1);
            lb[p] = id;
            if (id < maxc) {
                int* e = o + id * 8;
                e[0] = p; e[1] = W; e[2] = H; e[3] = -1; e[4] = -1; e[5] = 0; e[6] = 0; e[7] = 0;
            }
        }
    }
    __syncthreads();
    for (int p = tid; p < HW; p += CT_THREADS) {
        const int l = lf[p];
        if (l <= 0) continue;
        const int id = lb[l - 1];
        if (id >= maxc) continue;
        const int y = p / W, x = p - y * W;
        int* e = o + id * 8;
        atomicMin(&e[1], x); atomicMin(&e[2], y); atomicMax(&e[3], x); atomicMax(&e[4], y);
        atomicAdd(&e[5], 1);
        if (y + 1 < H && x + 1 < W) {              // the 2x2 cell whose top-left centre is this pixel
            const int k = 1 + (lf[p + 1] > 0) + (lf[p + W] > 0) + (lf[p + W + 1] > 0);
            if (k >= 3) atomicAdd(&e[6], k == 4 ? 2 : 1);
        }
    }
    // cells whose top-left corner is NOT in X but whose other three corners are
    for (int p = tid; p < HW; p += CT_THREADS) {
        if (lf[p] > 0) continue;
        const int y = p / W, x = p - y * W;
        if (y + 1 < H && x + 1 < W && lf[p + 1] > 0 && lf[p + W] > 0 && lf[p + W + 1] > 0) {
            const int id = lb[lf[p + 1] - 1];
            if (id < maxc) atomicAdd(&o[id * 8 + 6], 1);
        }
    }
    __syncthreads();
    if (tid == 0) count[img] = s_count;
}

extern "C" size_t advs_mask_contours_work_bytes(int n, int h, int w) { return (size_t)n * 2 * h * w * sizeof(int); }

extern "C" int advs_mask_contours(const uint8_t* mask, int n, int h, int w, void* work, int* out, int* count, int max_components,
                                  void* stream) {
    ADVS_REQUIRE(mask && work && out && count, "mask_contours: null pointer");
    ADVS_REQUIRE(n > 0 && h > 0 && w > 0 && (long long)h * w < (1ll << 30) && max_components > 0, "mask_contours: bad shape n=%d h=%d w=%d", n, h, w);
    mask_contours_kernel<<<n, CT_THREADS, 0, (hipStream_t)stream>>>(mask, h, w, (int*)work, out, count, max_components);
    ADVS_CHECK_LAUNCH("mask_contours");
    return ADVS_OK;
}
