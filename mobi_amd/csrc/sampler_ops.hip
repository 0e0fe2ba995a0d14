// Sampler-side fp32 arithmetic on the latent state (tiny, launch-bound; captured in a
// HIP graph together with the UNet forward by the host).
#include "common.h"

namespace mobi {

__global__ void ddim_step_kernel(const mobi_ddim_step_params a) {
  // all coefficient math in fp32, in the reference's operation order (ddim.py:200-212)
  float a_t = a.a_t, a_prev = a.a_prev, sigma_t = a.sigma_t, sqrt_one_minus_at = a.sqrt_one_minus_at;
  if (a.coef_dev) {           // graph-captured step: coefficients of the current step live in device memory
    a_t = a.coef_dev[0]; a_prev = a.coef_dev[1]; sigma_t = a.coef_dev[2]; sqrt_one_minus_at = a.coef_dev[3];
  }
  const float sqrt_at = sqrtf(a_t);
  const float sqrt_aprev = sqrtf(a_prev);
  const float dir_c = sqrtf(1.0f - a_prev - sigma_t * sigma_t);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n;
       i += (long long)gridDim.x * blockDim.x) {
    float e = a.e_cond[i];
    if (a.e_uncond) {
      const float eu = a.e_uncond[i];
      e = eu + a.cfg_scale * (e - eu);
    }
    const float x = a.x[i];
    const float pred = (x - sqrt_one_minus_at * e) / sqrt_at;
    const float dir = dir_c * e;
    const float nz = a.noise ? sigma_t * a.noise[i] * a.temperature : 0.0f;
    if (a.e_out) a.e_out[i] = e;
    if (a.pred_x0) a.pred_x0[i] = pred;
    if (a.x_prev) a.x_prev[i] = sqrt_aprev * pred + dir + nz;
  }
}

__global__ void lincomb4_kernel(float* out, const float* e0, const float* e1, const float* e2, const float* e3,
                                float c0, float c1, float c2, float c3, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = c0 * e0[i];
    if (e1) v += c1 * e1[i];
    if (e2) v += c2 * e2[i];
    if (e3) v += c3 * e3[i];
    out[i] = v;
  }
}

// q_sample (ddpm.py:284-287 of the reference): per-image gather of the two schedule tables by the int64 timestep
__global__ void q_sample_kernel(const float* x0, const float* noise, const long long* t, const float* sqrt_ac,
                                const float* sqrt_1m_ac, float* out, int batch, int per_image, int table_len) {
  const long long total = (long long)batch * per_image;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_image);
    long long ti = t[b];
    ti = ti < 0 ? 0 : (ti >= table_len ? table_len - 1 : ti);      // (torch.gather would raise; never read out of range)
    out[i] = sqrt_ac[ti] * x0[i] + sqrt_1m_ac[ti] * noise[i];
  }
}

__global__ void mask_blend_kernel(float* img, const float* x0, const float* noise, const float* mask, float sa,
                                  float s1, int batch, int channels, int hw) {
  const long long total = (long long)batch * channels * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % hw);
    const int b = (int)(i / ((long long)channels * hw));
    const float m = mask[(long long)b * hw + p];
    const float q = sa * x0[i] + s1 * noise[i];
    img[i] = q * m + (1.0f - m) * img[i];
  }
}

// range-view post-processing (include/mobi_engine.h, mobi_range_denorm).  Same operation order as the reference's
// torch expressions; this file is compiled with -ffp-contract=off, so no FMA is formed.
__global__ void range_denorm_kernel(const float* sample, const float* min_d, const float* max_d, float alpha,
                                    float two_alpha, float alpha_m1, float one_m_alpha, int object_norm, int int_norm,
                                    float* depth_out, float* int_out, int batch, int hw) {
  const long long total = (long long)batch * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / hw);
    const int p = (int)(i - (long long)b * hw);
    if (depth_out) {
      const float x = sample[((long long)b * 2) * hw + p];
      float d = x;
      if (object_norm) {
        const float lo = min_d[b], hi = max_d[b];
        if (x >= -alpha && x <= alpha) d = lo + (x + alpha) * (hi - lo) / two_alpha;
        else if (x >= -1.0f && x < -alpha) d = -1.0f + -(x + 1.0f) * (lo + 1.0f) / alpha_m1;
        else if (x > alpha && x <= 1.0f) d = hi + (x - alpha) * (1.0f - hi) / one_m_alpha;
      }
      depth_out[i] = d;
    }
    if (int_out) {
      const float x = sample[((long long)b * 2 + 1) * hw + p];
      float v = x;
      if (int_norm) {
        v = -0.5f * logf(1.0f - (x + 1.0f) / 2.0f) - 1.0f;
        v = fminf(fmaxf(v, -1.0f), 1.0f);
      }
      int_out[i] = v;
    }
  }
}

__global__ void posterior_sample_kernel(const float* moments, const float* noise, float* out, int batch, int c, int hw,
                                        int out_c_total, int out_c_off, float scale) {
  const long long total = (long long)batch * c * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % hw);
    const long long r = i / hw;
    const int ch = (int)(r % c);
    const int b = (int)(r / c);
    const float mean = moments[((long long)b * 2 * c + ch) * hw + p];
    float logvar = moments[((long long)b * 2 * c + c + ch) * hw + p];
    logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);
    const float z = mean + expf(0.5f * logvar) * noise[i];
    out[((long long)b * out_c_total + out_c_off + ch) * hw + p] = scale * z;
  }
}

__global__ void nearest_resize_kernel(const float* src, float* out, int planes, int hin, int win, int hout, int wout,
                                      int out_plane_stride) {
  const long long total = (long long)planes * hout * wout;
  // PyTorch 'nearest': src = floor(dst * (in / out)) computed in fp32 (legacy nearest)
  const float sh = (float)hin / (float)hout, sw = (float)win / (float)wout;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % wout);
    const long long r = i / wout;
    const int y = (int)(r % hout);
    const int pl = (int)(r / hout);
    const int sy = min((int)floorf((float)y * sh), hin - 1);
    const int sx = min((int)floorf((float)x * sw), win - 1);
    out[(long long)pl * out_plane_stride + (long long)y * wout + x] = src[((long long)pl * hin + sy) * win + sx];
  }
}

static inline unsigned egrid(long long n) {
  long long g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace mobi

using namespace mobi;
#define ST(stream) reinterpret_cast<hipStream_t>(stream)

extern "C" int mobi_ddim_step(const mobi_ddim_step_params* p, void* stream) {
  if (!p || !p->x || !p->e_cond || p->n <= 0) return MOBI_ERR_ARG;
  if (p->sigma_t != 0.0f && !p->noise) return MOBI_ERR_ARG;      // the stochastic term cannot be dropped silently
  hipLaunchKernelGGL(ddim_step_kernel, dim3(egrid(p->n)), dim3(256), 0, ST(stream), *p);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_lincomb4(float* out, const float* e0, const float* e1, const float* e2, const float* e3, float c0,
                             float c1, float c2, float c3, int64_t n, void* stream) {
  if (!out || !e0 || n <= 0) return MOBI_ERR_ARG;
  hipLaunchKernelGGL(lincomb4_kernel, dim3(egrid(n)), dim3(256), 0, ST(stream), out, e0, e1, e2, e3, c0, c1, c2, c3,
                     (long long)n);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac,
                             const float* sqrt_1m_ac, float* out, int32_t batch, int32_t per_image, int32_t table_len,
                             void* stream) {
  if (!x0 || !noise || !t || !sqrt_ac || !sqrt_1m_ac || !out || batch <= 0 || per_image <= 0 || table_len <= 0)
    return MOBI_ERR_ARG;
  hipLaunchKernelGGL(q_sample_kernel, dim3(egrid((long long)batch * per_image)), dim3(256), 0, ST(stream), x0, noise,
                     reinterpret_cast<const long long*>(t), sqrt_ac, sqrt_1m_ac, out, batch, per_image, table_len);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_mask_blend(float* img, const float* x0, const float* noise, const float* mask, float sqrt_ac_t,
                               float sqrt_1m_ac_t, int32_t batch, int32_t channels, int32_t hw, void* stream) {
  if (!img || !x0 || !noise || !mask || batch <= 0 || channels <= 0 || hw <= 0) return MOBI_ERR_ARG;
  hipLaunchKernelGGL(mask_blend_kernel, dim3(egrid((long long)batch * channels * hw)), dim3(256), 0, ST(stream), img,
                     x0, noise, mask, sqrt_ac_t, sqrt_1m_ac_t, batch, channels, hw);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_posterior_sample(const float* moments, const float* noise, float* out, int32_t batch, int32_t c,
                                     int32_t hw, int32_t out_c_total, int32_t out_c_off, float scale, void* stream) {
  if (!moments || !noise || !out || batch <= 0 || c <= 0 || hw <= 0 || out_c_off < 0 || out_c_off + c > out_c_total)
    return MOBI_ERR_ARG;
  hipLaunchKernelGGL(posterior_sample_kernel, dim3(egrid((long long)batch * c * hw)), dim3(256), 0, ST(stream),
                     moments, noise, out, batch, c, hw, out_c_total, out_c_off, scale);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_range_denorm(const float* sample, const float* min_d, const float* max_d, float alpha, float two_alpha,
                                 float alpha_m1, float one_m_alpha, int32_t object_norm, int32_t int_norm,
                                 float* depth_out, float* int_out, int32_t batch, int32_t hw, void* stream) {
  if (!sample || (!depth_out && !int_out) || batch <= 0 || hw <= 0) return MOBI_ERR_ARG;
  if (object_norm && depth_out && (!min_d || !max_d || !(alpha > 0.0f) || alpha > 1.0f)) return MOBI_ERR_ARG;
  hipLaunchKernelGGL(range_denorm_kernel, dim3(egrid((long long)batch * hw)), dim3(256), 0, ST(stream), sample, min_d,
                     max_d, alpha, two_alpha, alpha_m1, one_m_alpha, object_norm, int_norm, depth_out, int_out, batch, hw);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_nearest_resize(const float* src, float* out, int32_t planes, int32_t hin, int32_t win, int32_t hout,
                                   int32_t wout, int32_t out_plane_stride, void* stream) {
  if (!src || !out || planes <= 0 || hin <= 0 || win <= 0 || hout <= 0 || wout <= 0) return MOBI_ERR_ARG;
  hipLaunchKernelGGL(nearest_resize_kernel, dim3(egrid((long long)planes * hout * wout)), dim3(256), 0, ST(stream),
                     src, out, planes, hin, win, hout, wout, out_plane_stride);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_abi_version(void) { return MOBI_ABI_VERSION; }

extern "C" size_t mobi_struct_size(int id) {
  switch (id) {
    case 0: return sizeof(mobi_igemm_params);
    case 1: return sizeof(mobi_groupnorm_params);
    case 2: return sizeof(mobi_layernorm_params);
    case 3: return sizeof(mobi_attention_params);
    case 4: return sizeof(mobi_ctx_attention_params);
    case 5: return sizeof(mobi_skinny_linear_params);
    case 6: return sizeof(mobi_conv_small_cin_params);
    case 7: return sizeof(mobi_conv_small_cout_params);
    case 8: return sizeof(mobi_ddim_step_params);
    case 9: return sizeof(mobi_two_key_adapter_params);
    case 10: return sizeof(mobi_range_paste_params);
    case 11: return sizeof(mobi_lidar_metrics_params);
    case 12: return sizeof(mobi_range_prepare_params);
    case 13: return sizeof(mobi_image_prepare_params);
    case 14: return sizeof(mobi_ff_geglu_params);
    case 15: return sizeof(mobi_row_chain_params);
    case 16: return sizeof(mobi_chain_op);
    case 17: return sizeof(mobi_layernorm_bwd_params);
    case 18: return sizeof(mobi_attention_bwd_params);
    case 19: return sizeof(mobi_split_source);
    default: return 0;
  }
}

extern "C" const char* mobi_error_string(int code) {
  switch (code) {
    case MOBI_OK: return "ok";
    case MOBI_ERR_ARG: return "invalid argument";
    case MOBI_ERR_UNSUPPORTED: return "unsupported shape or mode";
    case MOBI_ERR_LAUNCH: return "kernel launch failed";
    case MOBI_ERR_ALIGN: return "pointer or stride not 16-byte aligned";
    default: return "unknown error";
  }
}
