// moog_raster.hip -- the rasteriser kernel (see moog_raster.h for the design), its own
// translation unit so that it builds independently of the step / reset kernels.
#include "moog_raster_kernel.h"
#include "moog_raster_mask.h"

template <int WORDS>
__global__ __launch_bounds__(R_THREADS, 6) void moog_raster_kernel(RArgs a) { raster_block<WORDS>(a, (int)blockIdx.x, -1); }

int moog_raster_configure_mask(size_t lds_bytes) { return moog_raster_mask_configure(lds_bytes); }

int moog_raster_configure(size_t lds_bytes) {
  hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(moog_raster_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (err == hipSuccess)
    err = hipFuncSetAttribute(reinterpret_cast<const void*>(moog_raster_kernel<2>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  return (int)err;
}

static RmArgs mask_args(const RArgs& r) {
  RmArgs a;
  memset(&a, 0, sizeof a);
  a.draw = r.em.out; a.lay = r.em.lay; a.image = r.image;
  a.n_envs = r.n_envs; a.S = r.ms.S; a.big = r.ms.big; a.compact = r.ms.compact; a.cap_rows = r.ms.cap_rows; a.W = r.canvas_w; a.H = r.canvas_h;
  a.flip = r.flip; a.iwords = r.ms.iwords; a.bg = r.ms.bg; a.debug_stop = r.debug_stop; a.threads = RM_THREADS;
  a.n_static = r.em.ncopy > 1 ? 0 : r.em.n_static;   // (torus frames are drawn whole: no cached picture under copies)
  a.sbg = r.sbg; a.rows_seen = r.rows_seen;
  a.plan = r.ms.plan;
  return a;
}

// Ordinary frames of programs the mask rasteriser takes (RmSetup::ok) are drawn by it; the pictures of the static / per-env
// prefix, frames on top of a per-env prefix and every other program's frames by the push / sort / span kernel.
void moog_raster_launch(const RArgs& a, size_t lds_bytes, hipStream_t stream) {
  if (a.ms.ok && !a.build && a.sbg_env_stride == 0 && !a.env_build) {
    if (!a.draw_ready) {   // records the engine did not step itself: the emitter on the records in HBM
      RmDeriveArgs d;
      d.em = a.em; d.P = a.P; d.L = a.L; d.f64 = a.f64; d.i32 = a.i32; d.vinfo = a.vinfo; d.n_envs = a.n_envs; d.env0 = a.env0;
      moog_draw_derive_launch(d, stream);
    }
    moog_raster_mask_launch(mask_args(a), a.ms.lds, stream, a.ms.persist_slots);
    return;
  }
  const dim3 grid((unsigned)a.n_envs * (unsigned)(a.tiles_x * a.bands));
  if (a.words > 1) hipLaunchKernelGGL(moog_raster_kernel<2>, grid, dim3(R_THREADS), lds_bytes, stream, a);
  else hipLaunchKernelGGL(moog_raster_kernel<1>, grid, dim3(R_THREADS), lds_bytes, stream, a);
}

// ---- Image.resize(size, resample=LANCZOS) (pil_renderer.py:112; Pillow Resample.c, 8 bits per channel): a horizontal
// pass, then a vertical pass, each rounding to uint8 (clip8); the fixed-point coefficient tables come from the host
// (moog_engine.hip resize_coeffs, the oracle's oracle_resize_coeffs restated).  One thread per output pixel.
#define R_PRECISION_BITS (32 - 8 - 2)
__device__ __forceinline__ unsigned r_clip8(int v) {
  v >>= R_PRECISION_BITS;
  return (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// Pillow's two resampling passes (Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc), fixed-point as there.
// Horizontal: a block = 256 consecutive output columns of one canvas row; the bytes they read (a contiguous span of
// the row) are staged in LDS with dword loads, the taps then come from LDS.  grid: x = column blocks, y / z = rows.
__global__ __launch_bounds__(256) void moog_resize_h_kernel(RResize r, const uint8_t* in, uint8_t* tmp, int n) {
  extern __shared__ uint8_t stage[];
  const size_t row = (size_t)blockIdx.y + (size_t)blockIdx.z * 65535u;   // env * ch + y
  if (row >= (size_t)n * r.ch) return;
  const int x0 = (int)blockIdx.x * 256, xx = x0 + (int)threadIdx.x;
  const int xl = (x0 + 255 < r.ow) ? x0 + 255 : r.ow - 1;
  const int b0 = (3 * r.bh[2 * x0]) & ~3, b1 = 3 * (r.bh[2 * xl] + r.bh[2 * xl + 1]);
  const uint8_t* rowp = in + row * (size_t)r.cstride * 3;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(rowp + b0);
  uint32_t* dst = reinterpret_cast<uint32_t*>(stage);
  const int ndw = (b1 - b0 + 3) >> 2;
  const int row_dw = (r.cstride * 3 - b0) >> 2;   // dwords left in the row (its length in memory is a multiple of 4)
  for (int t = threadIdx.x; t < ndw; t += 256) dst[t] = t < row_dw ? src[t] : 0u;
  __syncthreads();
  if (xx >= r.ow) return;
  const int xmin = r.bh[2 * xx], xmax = r.bh[2 * xx + 1];
  const int32_t* k = r.ch_coef + (size_t)xx * r.kh;
  const uint8_t* p = stage + (3 * xmin - b0);
  int s0 = 1 << (R_PRECISION_BITS - 1), s1 = s0, s2 = s0;
  for (int x = 0; x < xmax; ++x) { const int kk = k[x]; s0 += p[3 * x] * kk; s1 += p[3 * x + 1] * kk; s2 += p[3 * x + 2] * kk; }
  uint8_t* o = tmp + (row * r.tstride + xx) * 3;
  o[0] = (uint8_t)r_clip8(s0); o[1] = (uint8_t)r_clip8(s1); o[2] = (uint8_t)r_clip8(s2);
}

// Vertical: every byte of an output row is the same weighted sum over the rows above it, whatever its channel: a
// thread takes four consecutive bytes (one dword load per tap).  grid: x = dword blocks of a row, y = output row, z = env.
__global__ __launch_bounds__(256) void moog_resize_v_kernel(RResize r, const uint8_t* tmp, uint8_t* out, int n) {
  const int d = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int row_dw = (r.tstride * 3) >> 2;
  if (d >= row_dw) return;
  const int yy = (int)blockIdx.y;
  const size_t env = blockIdx.z;
  const int ymin = r.bv[2 * yy], ymax = r.bv[2 * yy + 1];
  const int32_t* k = r.cv_coef + (size_t)yy * r.kv;
  const uint8_t* p = tmp + (env * r.ch + ymin) * (size_t)r.tstride * 3 + 4 * (size_t)d;
  int s0 = 1 << (R_PRECISION_BITS - 1), s1 = s0, s2 = s0, s3 = s0;
  for (int y = 0; y < ymax; ++y) {
    const int kk = k[y];
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p + (size_t)y * r.tstride * 3);
    s0 += (int)(w & 255u) * kk; s1 += (int)((w >> 8) & 255u) * kk; s2 += (int)((w >> 16) & 255u) * kk; s3 += (int)(w >> 24) * kk;
  }
  uint8_t* o = out + (env * r.oh + (r.oh - 1 - yy)) * (size_t)r.ow * 3;   // (rows of the frame need not be dword aligned)
  const int b = 4 * d, nb = r.ow * 3;
  if (b < nb) o[b] = (uint8_t)r_clip8(s0);
  if (b + 1 < nb) o[b + 1] = (uint8_t)r_clip8(s1);
  if (b + 2 < nb) o[b + 2] = (uint8_t)r_clip8(s2);
  if (b + 3 < nb) o[b + 3] = (uint8_t)r_clip8(s3);
}

__global__ __launch_bounds__(256) void moog_crop_kernel(const uint8_t* in, uint8_t* out, size_t rows, int in_stride, int row_bytes) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint8_t* src = in + row * (size_t)in_stride;
  uint8_t* dst = out + row * (size_t)row_bytes;
  for (int b = threadIdx.x & 63; b < row_bytes; b += 64) dst[b] = src[b];
}

__global__ __launch_bounds__(64) void moog_prefix_check_kernel(PCArgs a) {
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= a.n_envs) return;
  PProg P = as_const_prog(a.P);
  const moog_layout_t& L = a.L;
  const double* gf = a.f64 + (size_t)env * L.f64_per_env;
  const int32_t* gq = a.i32 + (size_t)env * L.i32_per_env;
  double* sf = a.s_f64 + (size_t)env * L.f64_per_env;
  int32_t* sq = a.s_i32 + (size_t)env * L.i32_per_env;
  const int had = a.valid[env];
  int first_bad = a.n_static;
  if (had) {
    for (int s = lane; s < a.n_static; s += 64) {
      const int fl = gq[L.o_flags + s], nv = gq[L.o_nverts + s];
      bool bad = ((fl ^ sq[L.o_flags + s]) & MOOG_F_ALIVE) != 0;
      if (fl & MOOG_F_ALIVE) {
        bad = bad || nv != sq[L.o_nverts + s] || gq[L.o_opacity + s] != sq[L.o_opacity + s];
        for (int k = 0; k < 3; ++k)
          bad = bad || __double_as_longlong(gf[L.o_color + 3 * s + k]) != __double_as_longlong(sf[L.o_color + 3 * s + k]);
        const int v0 = L.o_verts + 2 * P->slot_voff[s];
        const int n2 = 2 * (nv < P->slot_vcap[s] ? nv : P->slot_vcap[s]);
        for (int k = 0; k < n2; ++k) bad = bad || __double_as_longlong(gf[v0 + k]) != __double_as_longlong(sf[v0 + k]);
      }
      if (bad && s < first_bad) first_bad = s;
    }
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(first_bad, o); first_bad = other < first_bad ? other : first_bad; }
  }
  const bool rebuild = !had || first_bad < a.n_static;
  if (rebuild) {
    const double2* src = reinterpret_cast<const double2*>(gf);
    double2* dst = reinterpret_cast<double2*>(sf);
    for (int i = lane; i < L.f64_per_env / 2; i += 64) dst[i] = src[i];
    const int4* srci = reinterpret_cast<const int4*>(gq);
    int4* dsti = reinterpret_cast<int4*>(sq);
    for (int i = lane; i < L.i32_per_env / 4; i += 64) dsti[i] = srci[i];
  }
  if (lane == 0) {
    a.build[env] = rebuild ? 1 : 0;
    a.valid[env] = 1;
    if (had && first_bad < a.n_static && gq[L.o_step_count] != 0)
      __hip_atomic_fetch_min(a.min_changed, first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

void moog_prefix_check_launch(const PCArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(moog_prefix_check_kernel, dim3((unsigned)a.n_envs), dim3(64), 0, stream, a);
}

void moog_crop_launch(const uint8_t* in, uint8_t* out, size_t rows, int in_stride, int row_bytes, hipStream_t stream) {
  hipLaunchKernelGGL(moog_crop_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, in, out, rows, in_stride, row_bytes);
}

void moog_resize_launch(const RResize& r, const uint8_t* canvas, uint8_t* tmp, uint8_t* out, int n, hipStream_t stream) {
  const unsigned bx = (unsigned)((r.ow + 255) / 256);
  const size_t rows = (size_t)n * r.ch;   // (grid.y is limited to 65535: the rest goes to z)
  const unsigned gy = (unsigned)(rows < 65535u ? rows : 65535u), gz = (unsigned)((rows + 65534u) / 65535u);
  hipLaunchKernelGGL(moog_resize_h_kernel, dim3(bx, gy, gz), dim3(256), (size_t)r.hspan, stream, r, canvas, tmp, n);
  const unsigned bd = (unsigned)(((r.tstride * 3 >> 2) + 255) / 256);
  for (int e0 = 0; e0 < n; e0 += 65535) {   // (grid.z likewise)
    const int m = n - e0 < 65535 ? n - e0 : 65535;
    hipLaunchKernelGGL(moog_resize_v_kernel, dim3(bd, (unsigned)r.oh, (unsigned)m), dim3(256), 0, stream, r,
                       tmp + (size_t)e0 * r.ch * r.tstride * 3, out + (size_t)e0 * r.oh * r.ow * 3, m);
  }
}
