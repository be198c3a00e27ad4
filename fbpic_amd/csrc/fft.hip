// Batched 1-D complex128 FFT along z through rocFFT, operating directly on the strided
// (Nz, ncols) view of one or several side-by-side (Nz, Nr) grids: element (iz, col) at
// base + iz*stride + col.  The reference's cuFFT path needs two transpose-copy kernels
// and a scaling kernel per transform (fbpic/fields/spectral_transform/fourier.py:116-160);
// here the column batch is described to rocFFT (stride = row stride, distance = 1) and the
// 1/Nz of the backward transform is folded into the plan (rocFFT scale factor).
#include "fb_common.h"
#include <rocfft/rocfft.h>
#include <mutex>

namespace fb {

struct FftPlan {
    rocfft_plan fwd = nullptr, bwd = nullptr;
    rocfft_execution_info info_f = nullptr, info_b = nullptr;
    void *work = nullptr;
    size_t work_bytes = 0;
    int inplace = 0;
};

static std::once_flag g_rocfft_once;

static int rfail(rocfft_status st, const char *where)
{
    if (st == rocfft_status_success) return 0;
    char buf[64];
    snprintf(buf, sizeof(buf), "rocfft_status %d", (int)st);
    set_error(where, buf);
    return 1000 + (int)st;
}

static int make_one(rocfft_plan *plan, rocfft_transform_type type, int Nz, long ncols,
                    long in_stride, long out_stride, int inplace, double scale)
{
    rocfft_plan_description desc = nullptr;
    int r = rfail(rocfft_plan_description_create(&desc), "fb_fft_plan_create(desc)");
    if (r) return r;
    size_t is[1] = {(size_t)in_stride}, os[1] = {(size_t)out_stride};
    r = rfail(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved,
                  rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, is, 1, 1, os, 1),
              "fb_fft_plan_create(layout)");
    if (!r && scale != 1.0)
        r = rfail(rocfft_plan_description_set_scale_factor(desc, scale), "fb_fft_plan_create(scale)");
    if (!r) {
        size_t len[1] = {(size_t)Nz};
        r = rfail(rocfft_plan_create(plan, inplace ? rocfft_placement_inplace : rocfft_placement_notinplace,
                                     type, rocfft_precision_double, 1, len, (size_t)ncols, desc),
                  "fb_fft_plan_create(plan)");
    }
    rocfft_plan_description_destroy(desc);
    return r;
}

}  // namespace fb

using namespace fb;

extern "C" int fb_fft_plan_create(int Nz, long ncols, long in_stride, long out_stride, int inplace,
                                  void **out_plan)
{
    std::call_once(g_rocfft_once, [] { rocfft_setup(); });
    if (inplace && in_stride != out_stride) {
        set_error("fb_fft_plan_create", "in-place plan needs equal strides");
        return -1;
    }
    FftPlan *p = new FftPlan();
    p->inplace = inplace;
    int r = make_one(&p->fwd, rocfft_transform_type_complex_forward, Nz, ncols, in_stride,
                     out_stride, inplace, 1.0);
    if (!r) r = make_one(&p->bwd, rocfft_transform_type_complex_inverse, Nz, ncols, in_stride,
                         out_stride, inplace, 1.0 / (double)Nz);
    size_t wf = 0, wb = 0;
    if (!r) r = rfail(rocfft_plan_get_work_buffer_size(p->fwd, &wf), "fb_fft_plan_create(work)");
    if (!r) r = rfail(rocfft_plan_get_work_buffer_size(p->bwd, &wb), "fb_fft_plan_create(work)");
    if (!r) {
        p->work_bytes = wf > wb ? wf : wb;
        if (p->work_bytes) {
            hipError_t e = hipMalloc(&p->work, p->work_bytes);
            if (e != hipSuccess) r = check(e, "fb_fft_plan_create(hipMalloc)");
        }
    }
    if (!r) r = rfail(rocfft_execution_info_create(&p->info_f), "fb_fft_plan_create(info)");
    if (!r) r = rfail(rocfft_execution_info_create(&p->info_b), "fb_fft_plan_create(info)");
    if (!r && p->work_bytes) {
        r = rfail(rocfft_execution_info_set_work_buffer(p->info_f, p->work, p->work_bytes), "fb_fft(work)");
        if (!r) r = rfail(rocfft_execution_info_set_work_buffer(p->info_b, p->work, p->work_bytes), "fb_fft(work)");
    }
    if (r) { fb_fft_plan_destroy(p); return r; }
    *out_plan = p;
    return 0;
}

extern "C" int fb_fft_exec(void *plan, int direction, const void *in, void *out, void *stream)
{
    FftPlan *p = (FftPlan *)plan;
    rocfft_execution_info info = direction < 0 ? p->info_f : p->info_b;
    int r = rfail(rocfft_execution_info_set_stream(info, stream), "fb_fft_exec(stream)");
    if (r) return r;
    void *ib[1] = {(void *)in};
    void *ob[1] = {out};
    return rfail(rocfft_execute(direction < 0 ? p->fwd : p->bwd, ib, p->inplace ? nullptr : ob, info),
                 "fb_fft_exec");
}

extern "C" int fb_fft_plan_destroy(void *plan)
{
    FftPlan *p = (FftPlan *)plan;
    if (!p) return 0;
    if (p->fwd) rocfft_plan_destroy(p->fwd);
    if (p->bwd) rocfft_plan_destroy(p->bwd);
    if (p->info_f) rocfft_execution_info_destroy(p->info_f);
    if (p->info_b) rocfft_execution_info_destroy(p->info_b);
    if (p->work) (void)hipFree(p->work);
    delete p;
    return 0;
}
