// Host side of the PIL-exact bilinear resize (fmgan_resize_* in include/fmgan_hip.h): the per-axis coefficient tables.
//
// The reference's loader runs transforms.Resize(size) on PIL images (train_3_encoder.py:233-239), i.e. Pillow's
// ImagingResample with the triangle filter (pinned pillow=8.2.0; src/libImaging/Resample.c).  Pillow derives, per
// output pixel, a tap window and double-precision weights, converts them to 22-bit fixed point and runs two integer
// passes.  The integer passes are the GPU kernel (image_io.hip); this file builds the tables in double on the host,
// once per (input size, output size).  It is compiled with -ffp-contract=off: an fma in `center = (xx + 0.5) * scale`
// or in the weight normalisation could move a value across a rounding boundary of the (int) conversions below.
#include <cmath>
#include <vector>

#include "fmgan_hip.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;   // fixed-point fraction of an 8-bit channel accumulated in int32
constexpr int kTileRows = 8;                 // output rows per block of the kernel (keep equal to RZ_TY in image_io.hip)

struct Axis {
  int ksize = 0;
  std::vector<int> bounds;   // [out][2]: first tap, tap count
  std::vector<int> coeffs;   // [out][ksize], unused taps 0
};

double triangle(double x) {
  if (x < 0.0) x = -x;
  return x < 1.0 ? 1.0 - x : 0.0;
}

Axis build_axis(int in_size, int out_size) {
  Axis a;
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;      // antialias only when shrinking
  const double support = 1.0 * filterscale;                   // triangle filter: support 1
  a.ksize = (int)std::ceil(support) * 2 + 1;
  a.bounds.assign(2 * (size_t)out_size, 0);
  a.coeffs.assign((size_t)a.ksize * out_size, 0);
  std::vector<double> w(a.ksize);
  const double inv = 1.0 / filterscale;
  for (int i = 0; i < out_size; ++i) {
    const double center = (i + 0.5) * scale;
    int first = (int)(center - support + 0.5);
    if (first < 0) first = 0;
    int last = (int)(center + support + 0.5);
    if (last > in_size) last = in_size;
    const int n = last - first;
    double total = 0.0;
    for (int t = 0; t < n; ++t) {
      w[t] = triangle((t + first - center + 0.5) * inv);
      total += w[t];
    }
    for (int t = 0; t < n; ++t) {
      double v = w[t];
      if (total != 0.0) v /= total;
      a.coeffs[(size_t)i * a.ksize + t] = v < 0 ? (int)(-0.5 + v * (1 << kPrecisionBits)) : (int)(0.5 + v * (1 << kPrecisionBits));
    }
    a.bounds[2 * i] = first;
    a.bounds[2 * i + 1] = n;
  }
  return a;
}

int max_row_span(const Axis& y, int out_h) {
  int span = 1;
  for (int y0 = 0; y0 < out_h; y0 += kTileRows) {
    const int yl = (y0 + kTileRows < out_h ? y0 + kTileRows : out_h) - 1;
    const int s = y.bounds[2 * yl] + y.bounds[2 * yl + 1] - y.bounds[2 * y0];
    if (s > span) span = s;
  }
  return span;
}

bool sizes_ok(int in_h, int in_w, int out_h, int out_w) {
  return in_h > 0 && in_w > 0 && out_h > 0 && out_w > 0 && in_h <= (1 << 20) && in_w <= (1 << 20) && out_h <= (1 << 20) &&
         out_w <= (1 << 20);
}

}  // namespace

extern "C" int fmgan_resize_output_size(int h, int w, int size, int* out_h, int* out_w) {
  if (h <= 0 || w <= 0 || size <= 0 || !out_h || !out_w) return FMGAN_EINVAL;
  const int shorter = w <= h ? w : h, longer = w <= h ? h : w;
  if (shorter == size) { *out_h = h; *out_w = w; return FMGAN_OK; }
  const int new_long = (int)((double)size * longer / shorter);   // Python: int(size * long / short), true division
  if (w <= h) { *out_w = size; *out_h = new_long; } else { *out_h = size; *out_w = new_long; }
  return (*out_h > 0 && *out_w > 0) ? FMGAN_OK : FMGAN_EINVAL;
}

extern "C" long long fmgan_resize_plan_ints(int in_h, int in_w, int out_h, int out_w) {
  if (!sizes_ok(in_h, in_w, out_h, out_w)) return 0;
  const double sx = (double)in_w / out_w, sy = (double)in_h / out_h;
  const long long kx = (long long)std::ceil(sx < 1.0 ? 1.0 : sx) * 2 + 1, ky = (long long)std::ceil(sy < 1.0 ? 1.0 : sy) * 2 + 1;
  return 8 + 2LL * out_w + kx * out_w + 2LL * out_h + ky * out_h;
}

extern "C" int fmgan_resize_plan(int in_h, int in_w, int out_h, int out_w, int* plan, long long plan_ints) {
  if (!sizes_ok(in_h, in_w, out_h, out_w) || !plan) return FMGAN_EINVAL;
  if (plan_ints < fmgan_resize_plan_ints(in_h, in_w, out_h, out_w)) return FMGAN_EINVAL;
  const Axis x = build_axis(in_w, out_w), y = build_axis(in_h, out_h);
  int* p = plan;
  p[0] = x.ksize; p[1] = y.ksize; p[2] = in_h; p[3] = in_w; p[4] = out_h; p[5] = out_w;
  p[6] = max_row_span(y, out_h); p[7] = kTileRows;
  p += 8;
  for (int v : x.bounds) *p++ = v;
  for (int v : x.coeffs) *p++ = v;
  for (int v : y.bounds) *p++ = v;
  for (int v : y.coeffs) *p++ = v;
  return FMGAN_OK;
}
