// Input front end: decoded RGB image, uint8 HWC  ->  Resize(size, bicubic) ->
// CenterCrop(size) -> ToTensor (float32 CHW in [0,1], NOT normalised: Normalize lives inside the model), i.e. the
// transform the reference applies on 8 CPU workers (train/adversarial_training_clip.py:105-116; torchvision 0.15.2
// Resize / CenterCrop / ToTensor over Pillow's resampler).
//
// Bit-exact with Pillow: the separable, antialiased bicubic (Keys a = -0.5, support 2 x down-scale factor) is
// evaluated exactly like ImagingResample does for 8-bit images - taps normalised in double precision and converted to
// 22-bit fixed point ON THE HOST (rvlm_preproc computes only the rows / columns the centre crop keeps), horizontal
// pass then vertical pass, each accumulated in int32 with the rounding constant 2^21, shifted and clipped to uint8.
// One kernel: a workgroup owns a TH x 32 tile of output pixels; the horizontal pass of the input rows that tile needs
// goes to LDS as uint8, the vertical pass reads it back and writes float32 / 255 (IEEE division, as torch does).
// HBM-bound byte work: H*W*3 bytes in (rows outside the crop are never read), size^2 * 12 bytes out.
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

#include "kernels.h"

namespace rvlm {

constexpr int PP_BITS = 32 - 8 - 2;   // Pillow's PRECISION_BITS for 8-bit channels
constexpr int PP_TW = 32;             // output tile width
constexpr int PP_LDS = 60 * 1024;     // bytes of the horizontal-pass tile

struct PreprocArgs {
    const uint8_t* img; int H, W;            // input [H, W, 3]
    int size;                                // output [3, size, size]
    const int* hb; const int* hk; int hks;   // horizontal: bounds [size][2], taps [size][hks]
    const int* vb; const int* vk; int vks;   // vertical
    int th;                                  // output rows per tile
    float* out;
};

__device__ __forceinline__ int clip8(int v) { v >>= PP_BITS; return v < 0 ? 0 : (v > 255 ? 255 : v); }

// one image of a batch: where its tap tables start in the batch's table buffer (int offsets)
struct PreprocImg {
    const uint8_t* img; int H, W;
    int hb, hk, hks, vb, vk, vks, th;
};

__device__ __forceinline__ void resize_crop_tile(const PreprocArgs& a, uint8_t* inter) {
    const int ox0 = blockIdx.x * PP_TW, oy0 = blockIdx.y * a.th;
    if (oy0 >= a.size) return;           // (batched launch: the grid is sized for the smallest tile height of the batch)
    const int tw = min(PP_TW, a.size - ox0), th = min(a.th, a.size - oy0);
    // input rows this tile's vertical taps touch
    const int r0 = a.vb[2 * oy0];
    const int r1 = a.vb[2 * (oy0 + th - 1)] + a.vb[2 * (oy0 + th - 1) + 1];
    const int rows = r1 - r0;
    // ---- horizontal pass: (row, ox, c) -> LDS ----
    for (int i = threadIdx.x; i < rows * tw * 3; i += 256) {
        const int c = i % 3, ox = (i / 3) % tw, r = i / (3 * tw);
        const int xmin = a.hb[2 * (ox0 + ox)], n = a.hb[2 * (ox0 + ox) + 1];
        const int* k = a.hk + (long)(ox0 + ox) * a.hks;
        const uint8_t* src = a.img + ((long)(r0 + r) * a.W + xmin) * 3 + c;
        int ss = 1 << (PP_BITS - 1);
        for (int t = 0; t < n; ++t) ss += (int)src[t * 3] * k[t];
        inter[(r * PP_TW + ox) * 3 + c] = (uint8_t)clip8(ss);
    }
    __syncthreads();
    // ---- vertical pass + ToTensor ----
    for (int i = threadIdx.x; i < th * tw * 3; i += 256) {
        const int ox = i % tw, oy = (i / tw) % th, c = i / (tw * th);
        const int ymin = a.vb[2 * (oy0 + oy)], n = a.vb[2 * (oy0 + oy) + 1];
        const int* k = a.vk + (long)(oy0 + oy) * a.vks;
        int ss = 1 << (PP_BITS - 1);
        for (int t = 0; t < n; ++t) ss += (int)inter[((ymin - r0 + t) * PP_TW + ox) * 3 + c] * k[t];
        a.out[((long)c * a.size + oy0 + oy) * a.size + ox0 + ox] = (float)clip8(ss) / 255.0f;
    }
}

__global__ void __launch_bounds__(256)
resize_crop_kernel(PreprocArgs a) {
    extern __shared__ uint8_t inter[];   // [rows][PP_TW][3]
    resize_crop_tile(a, inter);
}

// a whole batch in ONE launch: blockIdx.z = image, every image with its own shape, tables and tile height
__global__ void __launch_bounds__(256)
resize_crop_batch_kernel(const PreprocImg* __restrict__ imgs, const int* __restrict__ tables, int size, float* __restrict__ out) {
    extern __shared__ uint8_t inter[];
    const PreprocImg d = imgs[blockIdx.z];
    PreprocArgs a;
    a.img = d.img; a.H = d.H; a.W = d.W; a.size = size;
    a.hb = tables + d.hb; a.hk = tables + d.hk; a.hks = d.hks;
    a.vb = tables + d.vb; a.vk = tables + d.vk; a.vks = d.vks;
    a.th = d.th;
    a.out = out + (long)blockIdx.z * 3 * size * size;
    resize_crop_tile(a, inter);
}

// ---- host side: Pillow's precompute_coeffs + normalize_coeffs_8bpc for output indices [o0, o0 + count) ----
static double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}
static int resample_ksize(int in_size, int out_size) {
    double fs = (double)in_size / out_size;
    if (fs < 1.0) fs = 1.0;
    return (int)std::ceil(2.0 * fs) * 2 + 1;
}
static void resample_coeffs(int in_size, int out_size, int o0, int count, int ksize, std::vector<int>& bounds,
                            std::vector<int>& kk) {
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
    bounds.assign((size_t)count * 2, 0);
    kk.assign((size_t)count * ksize, 0);
    std::vector<double> w(ksize);
    for (int i = 0; i < count; ++i) {
        const double center = (o0 + i + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { w[x] = bicubic_filter((x + xmin - center + 0.5) * ss); ww += w[x]; }
        for (int x = 0; x < xmax; ++x) {
            const double k = (ww != 0.0) ? w[x] / ww : w[x];
            kk[(size_t)i * ksize + x] = k < 0 ? (int)(-0.5 + k * (1 << PP_BITS)) : (int)(0.5 + k * (1 << PP_BITS));
        }
        bounds[2 * i] = xmin;
        bounds[2 * i + 1] = xmax;
    }
}

}  // namespace rvlm

using namespace rvlm;

struct rvlm_preproc {
    int size = 0, max_ksize = 0;
    int *d_hb = nullptr, *d_hk = nullptr, *d_vb = nullptr, *d_vk = nullptr;
    int last_h = -1, last_w = -1, hks = 0, vks = 0, th = 0;   // tables on the device describe this input shape
    // batched form: descriptors + tap tables of one batch, staged in pinned host memory and copied asynchronously
    void* h_stage = nullptr; void* d_stage = nullptr; size_t stage_bytes = 0;
    hipEvent_t batch_done = nullptr;      // the last batch's kernel: its staging buffers are free once it has run
};

// resize geometry of torchvision Resize(int) + CenterCrop for one input shape, and its tap tables
struct PreprocPlan { int nh, nw, top, left, hks, vks, th; std::vector<int> hb, hk, vb, vk; };
static int preproc_plan(int H, int W, int size, int max_ksize, PreprocPlan& q) {
    // torchvision Resize(int): shorter edge -> size, the other edge int(size * long / short); CenterCrop offsets
    if (W <= H) { q.nw = size; q.nh = (W == H) ? size : (int)((double)size * H / W); }
    else { q.nh = size; q.nw = (int)((double)size * W / H); }
    q.top = (int)std::nearbyint((q.nh - size) / 2.0); q.left = (int)std::nearbyint((q.nw - size) / 2.0);
    q.hks = resample_ksize(W, q.nw); q.vks = resample_ksize(H, q.nh);
    if (q.hks > max_ksize || q.vks > max_ksize)
        return fail(RVLM_ERR_UNSUPPORTED, "rvlm_preproc: input larger than max_input_dim given at creation");
    resample_coeffs(W, q.nw, q.left, size, q.hks, q.hb, q.hk);
    resample_coeffs(H, q.nh, q.top, size, q.vks, q.vb, q.vk);
    // tile height: the horizontal-pass rows of a tile must fit the LDS budget
    int th = 32;
    for (; th >= 1; th >>= 1) {
        int worst = 0;
        for (int oy0 = 0; oy0 < size; oy0 += th) {
            const int l = std::min(oy0 + th, size) - 1;
            worst = std::max(worst, q.vb[2 * l] + q.vb[2 * l + 1] - q.vb[2 * oy0]);
        }
        if ((long)worst * PP_TW * 3 <= PP_LDS) break;
    }
    if (th < 1) return fail(RVLM_ERR_UNSUPPORTED, "rvlm_preproc: down-scaling factor too large");
    q.th = th;
    return RVLM_OK;
}

extern "C" int rvlm_preproc_create(int size, int max_input_dim, rvlm_preproc** out) {
    RVLM_REQUIRE(out && size > 0 && size <= 4096 && max_input_dim >= 1, "rvlm_preproc_create: bad arguments");
    auto* p = new rvlm_preproc;
    p->size = size;
    p->max_ksize = resample_ksize(std::max(max_input_dim, size), size);
    const size_t nb = (size_t)size * 2 * sizeof(int), nk = (size_t)size * p->max_ksize * sizeof(int);
    if (hipMalloc(&p->d_hb, nb) != hipSuccess || hipMalloc(&p->d_vb, nb) != hipSuccess ||
        hipMalloc(&p->d_hk, nk) != hipSuccess || hipMalloc(&p->d_vk, nk) != hipSuccess) {
        delete p;
        return fail(RVLM_ERR_HIP, "rvlm_preproc_create: hipMalloc failed");
    }
    *out = p;
    return RVLM_OK;
}

extern "C" int rvlm_preproc_destroy(rvlm_preproc* p) {
    if (!p) return RVLM_OK;
    (void)hipFree(p->d_hb); (void)hipFree(p->d_hk); (void)hipFree(p->d_vb); (void)hipFree(p->d_vk);
    if (p->batch_done) { (void)hipEventSynchronize(p->batch_done); (void)hipEventDestroy(p->batch_done); }
    if (p->h_stage) (void)hipHostFree(p->h_stage);
    if (p->d_stage) (void)hipFree(p->d_stage);
    delete p;
    return RVLM_OK;
}

extern "C" int rvlm_preproc_run(rvlm_preproc* p, const uint8_t* img_hwc, int H, int W, float* out_chw,
                                rvlm_stream_t stream) {
    RVLM_REQUIRE(p && img_hwc && out_chw && H > 0 && W > 0, "rvlm_preproc_run: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int size = p->size;
    if (H != p->last_h || W != p->last_w) {
        PreprocPlan q;
        if (int rc = preproc_plan(H, W, size, p->max_ksize, q)) return rc;
        const std::vector<int>&hb = q.hb, &hk = q.hk, &vb = q.vb, &vk = q.vk;
        const int hks = q.hks, vks = q.vks, th = q.th;
        // the previous image's kernel may still be reading the tables
        RVLM_HIP(hipStreamSynchronize(s));
        RVLM_HIP(hipMemcpy(p->d_hb, hb.data(), hb.size() * sizeof(int), hipMemcpyHostToDevice));
        RVLM_HIP(hipMemcpy(p->d_hk, hk.data(), hk.size() * sizeof(int), hipMemcpyHostToDevice));
        RVLM_HIP(hipMemcpy(p->d_vb, vb.data(), vb.size() * sizeof(int), hipMemcpyHostToDevice));
        RVLM_HIP(hipMemcpy(p->d_vk, vk.data(), vk.size() * sizeof(int), hipMemcpyHostToDevice));
        p->last_h = H; p->last_w = W; p->hks = hks; p->vks = vks; p->th = th;
    }
    PreprocArgs a;
    a.img = img_hwc; a.H = H; a.W = W; a.size = size;
    a.hb = p->d_hb; a.hk = p->d_hk; a.hks = p->hks;
    a.vb = p->d_vb; a.vk = p->d_vk; a.vks = p->vks;
    a.th = p->th; a.out = out_chw;
    static unsigned long long attr_devices = 0;
    RVLM_ONCE_PER_DEVICE(attr_devices, RVLM_HIP(hipFuncSetAttribute((const void*)resize_crop_kernel,
                                                                    hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS)));
    hipLaunchKernelGGL(resize_crop_kernel, dim3(cdiv(size, PP_TW), cdiv(size, p->th)), dim3(256), PP_LDS, s, a);
    RVLM_CHECK_LAUNCH();
    return RVLM_OK;
}

// The reference feeds batches of 128 images from 8 loader workers (train/adversarial_training_clip.py:119-148,
// train/datasets.py:38-47); here a batch of decoded images of ANY mix of shapes is one kernel launch: tap tables per
// distinct shape are computed on the host, staged with the per-image descriptors in pinned memory, copied
// asynchronously on the stream, and blockIdx.z walks the images.  No host synchronisation unless the previous
// batch's kernel has not run yet when its staging buffers are needed again.
extern "C" int rvlm_preproc_run_batch(rvlm_preproc* p, const uint8_t* const* imgs_hwc, const int* H, const int* W, int n,
                                      float* out, rvlm_stream_t stream) {
    RVLM_REQUIRE(p && imgs_hwc && H && W && out && n > 0 && n <= 65535, "rvlm_preproc_run_batch: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int size = p->size;
    std::vector<PreprocImg> desc((size_t)n);
    std::vector<int> tables;
    std::vector<std::pair<long, PreprocImg>> shapes;     // (H << 32 | W) -> offsets of that shape's tables
    int th_min = 32;
    for (int i = 0; i < n; ++i) {
        RVLM_REQUIRE(imgs_hwc[i] && H[i] > 0 && W[i] > 0, "rvlm_preproc_run_batch: bad image");
        const long key = ((long)H[i] << 32) | (unsigned)W[i];
        const PreprocImg* found = nullptr;
        for (auto& kv : shapes) if (kv.first == key) { found = &kv.second; break; }
        PreprocImg d;
        if (found) d = *found;
        else {
            PreprocPlan q;
            if (int rc = preproc_plan(H[i], W[i], size, p->max_ksize, q)) return rc;
            auto put = [&](const std::vector<int>& v) { const int o = (int)tables.size(); tables.insert(tables.end(), v.begin(), v.end()); return o; };
            d.hb = put(q.hb); d.hk = put(q.hk); d.vb = put(q.vb); d.vk = put(q.vk);
            d.hks = q.hks; d.vks = q.vks; d.th = q.th; d.H = H[i]; d.W = W[i];
            shapes.emplace_back(key, d);
        }
        d.img = imgs_hwc[i];
        desc[i] = d;
        th_min = std::min(th_min, d.th);
    }
    const size_t desc_bytes = (desc.size() * sizeof(PreprocImg) + 255) / 256 * 256, need = desc_bytes + tables.size() * sizeof(int);
    if (p->batch_done) RVLM_HIP(hipEventSynchronize(p->batch_done));     // (normally long done: a batch is microseconds)
    else RVLM_HIP(hipEventCreateWithFlags(&p->batch_done, hipEventDisableTiming));
    if (need > p->stage_bytes) {
        if (p->h_stage) (void)hipHostFree(p->h_stage);
        if (p->d_stage) (void)hipFree(p->d_stage);
        p->h_stage = nullptr; p->d_stage = nullptr;
        p->stage_bytes = need * 2;
        if (hipHostMalloc(&p->h_stage, p->stage_bytes, hipHostMallocDefault) != hipSuccess ||
            hipMalloc(&p->d_stage, p->stage_bytes) != hipSuccess) {
            p->stage_bytes = 0;
            return fail(RVLM_ERR_HIP, "rvlm_preproc_run_batch: staging allocation failed");
        }
    }
    memcpy(p->h_stage, desc.data(), desc.size() * sizeof(PreprocImg));
    memcpy((char*)p->h_stage + desc_bytes, tables.data(), tables.size() * sizeof(int));
    RVLM_HIP(hipMemcpyAsync(p->d_stage, p->h_stage, need, hipMemcpyHostToDevice, s));
    static unsigned long long attr_devices_b = 0;
    RVLM_ONCE_PER_DEVICE(attr_devices_b, RVLM_HIP(hipFuncSetAttribute((const void*)resize_crop_batch_kernel,
                                                                      hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS)));
    hipLaunchKernelGGL(resize_crop_batch_kernel, dim3(cdiv(size, PP_TW), cdiv(size, th_min), n), dim3(256), PP_LDS, s,
                       (const PreprocImg*)p->d_stage, (const int*)((const char*)p->d_stage + desc_bytes), size, out);
    RVLM_CHECK_LAUNCH();
    RVLM_HIP(hipEventRecord(p->batch_done, s));
    return RVLM_OK;
}
