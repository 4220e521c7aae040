/*
 * TEST INFRASTRUCTURE — CPU oracle for the LGDWT-GS loss terms.  Never on the product path.
 *
 * Restates (fp32, -ffp-contract=off; reductions accumulate in double):
 *   l1_loss                       /root/reference/fs3dgs_benchmark/LGDWT-GS/utils/loss_utils.py:40-41
 *   get_dwt_subbands              .../LGDWT-GS/utils/loss_utils.py:106-153
 *   global DWT loss               .../LGDWT-GS/train.py:132-164
 *   compute_elf_map               .../LGDWT-GS/utils/loss_utils.py:336-366
 *   compute_patch_dwt_loss        .../LGDWT-GS/utils/loss_utils.py:368-442
 *   fusedssim / fusedssim_backward  .../gaussian-splatting/submodules/fused-ssim/ssim.cu:187-366
 *
 * The Haar arithmetic itself lives in the third-party package pytorch_wavelets (DWTForward,
 * J=1, wave='db1', mode='symmetric'), which is NOT vendored in /root/reference, NOT version
 * pinned anywhere in it, and not installed here: **DWT parity is unpinned** by the reference.
 * What is restated is that package's published algorithm (pytorch_wavelets/dwt/lowlevel.py,
 * afb1d/AFB2D): analysis filters h0 = [c, c], h1 = [c, -c] with c = float32(0.7071067811865476)
 * (pywt 'db1' dec_lo / dec_hi, reversed by prep_filt_afb1d), correlation with stride 2 first along
 * W (dim 3) then along H (dim 2); for an odd length one sample is appended that repeats the last
 * one ('symmetric' pad p = 1 placed on the right); band order Yh[:, :, 0/1/2] = LH/HL/HH with
 * LH = low along W, high along H.
 * Pinned by: analytic KATs (constant / ramp / checkerboard images), orthonormality
 * (energy preservation, perfect reconstruction through the adjoint), torch autograd of a torch
 * restatement for the gradients, and for SSIM / L1 by golden values produced by the importable
 * reference python (tests/golden/make_golden.py).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/gsplat.h"

static const float HC = 0.7071067811865476f; /* float32(pywt db1 coefficient) */

static inline int cdiv2(int n) { return (n + 1) / 2; }
/* symmetric right pad by one: index n (only reachable for odd n) repeats n-1 */
static inline int symi(int i, int n) { return i < n ? i : n - 1; }

struct Bands {
  float ll, lh, hl, hh;
};
static inline Bands haar_block(float a, float b, float c, float d) {
  /* stage 1 along W */
  float lo_t = HC * a + HC * b, hi_t = HC * a - HC * b;
  float lo_b = HC * c + HC * d, hi_b = HC * c - HC * d;
  /* stage 2 along H */
  Bands r;
  r.ll = HC * lo_t + HC * lo_b;
  r.lh = HC * lo_t - HC * lo_b;
  r.hl = HC * hi_t + HC * hi_b;
  r.hh = HC * hi_t - HC * hi_b;
  return r;
}
/* adjoint of haar_block: gradients wrt a,b,c,d */
static inline void haar_block_adj(float gll, float glh, float ghl, float ghh, float& da, float& db,
                                  float& dc, float& dd) {
  float dlo_t = HC * gll + HC * glh, dlo_b = HC * gll - HC * glh;
  float dhi_t = HC * ghl + HC * ghh, dhi_b = HC * ghl - HC * ghh;
  da = HC * dlo_t + HC * dhi_t;
  db = HC * dlo_t - HC * dhi_t;
  dc = HC * dlo_b + HC * dhi_b;
  dd = HC * dlo_b - HC * dhi_b;
}
static inline float sgn(float x) { return (float)((x > 0.f) - (x < 0.f)); }

static void dwt_level(const float* x, int NC, int H, int W, float* ll, float* lh, float* hl, float* hh) {
  const int h = cdiv2(H), w = cdiv2(W);
  for (int c = 0; c < NC; c++)
    for (int i = 0; i < h; i++)
      for (int j = 0; j < w; j++) {
        const float* p = x + (size_t)c * H * W;
        int y0 = 2 * i, y1 = symi(2 * i + 1, H), x0 = 2 * j, x1 = symi(2 * j + 1, W);
        Bands b = haar_block(p[(size_t)y0 * W + x0], p[(size_t)y0 * W + x1], p[(size_t)y1 * W + x0],
                             p[(size_t)y1 * W + x1]);
        size_t o = ((size_t)c * h + i) * w + j;
        if (ll) ll[o] = b.ll;
        if (lh) lh[o] = b.lh;
        if (hl) hl[o] = b.hl;
        if (hh) hh[o] = b.hh;
      }
}
static void dwt_level_adj(const float* dll, const float* dlh, const float* dhl, const float* dhh, int NC,
                          int H, int W, float* dx, bool accumulate) {
  const int h = cdiv2(H), w = cdiv2(W);
  if (!accumulate) memset(dx, 0, sizeof(float) * (size_t)NC * H * W);
  for (int c = 0; c < NC; c++)
    for (int i = 0; i < h; i++)
      for (int j = 0; j < w; j++) {
        size_t o = ((size_t)c * h + i) * w + j;
        float da, db, dc, dd;
        haar_block_adj(dll ? dll[o] : 0.f, dlh ? dlh[o] : 0.f, dhl ? dhl[o] : 0.f, dhh ? dhh[o] : 0.f, da, db, dc, dd);
        float* p = dx + (size_t)c * H * W;
        int y0 = 2 * i, y1 = symi(2 * i + 1, H), x0 = 2 * j, x1 = symi(2 * j + 1, W);
        p[(size_t)y0 * W + x0] += da;
        p[(size_t)y0 * W + x1] += db;
        p[(size_t)y1 * W + x0] += dc;
        p[(size_t)y1 * W + x1] += dd;
      }
}

/* fused-ssim/ssim.cu:9-19 */
static const float GW[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f,
                             0.10936068743467331f,  0.21300552785396576f,  0.26601171493530273f,
                             0.21300552785396576f,  0.10936068743467331f,  0.036000773310661316f,
                             0.0075987582094967365f, 0.001028380123898387f};

/* separable 11-tap zero-padded convolution of one [H,W] plane: x pass then y pass, taps
 * accumulated in order 0..10 starting from 0 (ssim.cu:104-184) */
static void sepconv(const float* in, int H, int W, std::vector<float>& tmp, float* out) {
  tmp.assign((size_t)H * W, 0.f);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      float val = 0.f;
      for (int k = 0; k < 11; k++) {
        int xx = x + k - 5;
        float v = (xx >= 0 && xx < W) ? in[(size_t)y * W + xx] : 0.f;
        val += GW[k] * v;
      }
      tmp[(size_t)y * W + x] = val;
    }
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      float val = 0.f;
      for (int k = 0; k < 11; k++) {
        int yy = y + k - 5;
        float v = (yy >= 0 && yy < H) ? tmp[(size_t)yy * W + x] : 0.f;
        val += GW[k] * v;
      }
      out[(size_t)y * W + x] = val;
    }
}

extern "C" {

int gso_l1_fwd(const float* a, const float* b, int64_t n, float* sum, void*) {
  if (!a || !b || !sum) return GS_E_NULL;
  double s = 0;
  for (int64_t i = 0; i < n; i++) s += (double)fabsf(a[i] - b[i]);
  *sum += (float)s;
  return GS_OK;
}
int gso_l1_bwd(const float* a, const float* b, int64_t n, float coef, float* g, int32_t accumulate, void*) {
  if (!a || !b || !g) return GS_E_NULL;
  for (int64_t i = 0; i < n; i++) {
    float v = coef * sgn(a[i] - b[i]);
    g[i] = accumulate ? g[i] + v : v;
  }
  return GS_OK;
}

int gso_dwt_haar_fwd(const float* x, int32_t NC, int32_t H, int32_t W, float* ll, float* lh, float* hl,
                     float* hh, void*) {
  if (!x) return GS_E_NULL;
  dwt_level(x, NC, H, W, ll, lh, hl, hh);
  return GS_OK;
}
int gso_dwt_haar_bwd(const float* dll, const float* dlh, const float* dhl, const float* dhh, int32_t NC,
                     int32_t H, int32_t W, float* dx, void*) {
  if (!dx) return GS_E_NULL;
  dwt_level_adj(dll, dlh, dhl, dhh, NC, H, W, dx, false);
  return GS_OK;
}

int gso_dwt2_l1_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, float* band_sums,
                    void*) {
  if (!pred || !gt || !band_sums) return GS_E_NULL;
  const int h1 = cdiv2(H), w1 = cdiv2(W), h2 = cdiv2(h1), w2 = cdiv2(w1);
  const size_t n1 = (size_t)C * h1 * w1, n2 = (size_t)C * h2 * w2;
  std::vector<float> p1[4], g1[4], p2[4], g2[4];
  for (int k = 0; k < 4; k++) {
    p1[k].resize(n1); g1[k].resize(n1); p2[k].resize(n2); g2[k].resize(n2);
  }
  dwt_level(pred, C, H, W, p1[0].data(), p1[1].data(), p1[2].data(), p1[3].data());
  dwt_level(gt, C, H, W, g1[0].data(), g1[1].data(), g1[2].data(), g1[3].data());
  dwt_level(p1[0].data(), C, h1, w1, p2[0].data(), p2[1].data(), p2[2].data(), p2[3].data());
  dwt_level(g1[0].data(), C, h1, w1, g2[0].data(), g2[1].data(), g2[2].data(), g2[3].data());
  for (int k = 0; k < 4; k++) {
    double s = 0;
    for (size_t i = 0; i < n1; i++) s += (double)fabsf(p1[k][i] - g1[k][i]);
    band_sums[k] += (float)s;
    s = 0;
    for (size_t i = 0; i < n2; i++) s += (double)fabsf(p2[k][i] - g2[k][i]);
    band_sums[4 + k] += (float)s;
  }
  return GS_OK;
}

int gso_dwt2_l1_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, const float* coef,
                    float* grad_pred, int32_t accumulate, void*) {
  if (!pred || !gt || !coef || !grad_pred) return GS_E_NULL;
  const int h1 = cdiv2(H), w1 = cdiv2(W), h2 = cdiv2(h1), w2 = cdiv2(w1);
  const size_t n1 = (size_t)C * h1 * w1, n2 = (size_t)C * h2 * w2;
  std::vector<float> p1[4], g1[4], p2[4], g2[4];
  for (int k = 0; k < 4; k++) {
    p1[k].resize(n1); g1[k].resize(n1); p2[k].resize(n2); g2[k].resize(n2);
  }
  dwt_level(pred, C, H, W, p1[0].data(), p1[1].data(), p1[2].data(), p1[3].data());
  dwt_level(gt, C, H, W, g1[0].data(), g1[1].data(), g1[2].data(), g1[3].data());
  dwt_level(p1[0].data(), C, h1, w1, p2[0].data(), p2[1].data(), p2[2].data(), p2[3].data());
  dwt_level(g1[0].data(), C, h1, w1, g2[0].data(), g2[1].data(), g2[2].data(), g2[3].data());
  std::vector<float> d2[4], d1[4], dll1(n1);
  for (int k = 0; k < 4; k++) {
    d2[k].resize(n2);
    for (size_t i = 0; i < n2; i++) d2[k][i] = coef[4 + k] * sgn(p2[k][i] - g2[k][i]);
  }
  dwt_level_adj(d2[0].data(), d2[1].data(), d2[2].data(), d2[3].data(), C, h1, w1, dll1.data(), false);
  for (int k = 0; k < 4; k++) {
    d1[k].resize(n1);
    for (size_t i = 0; i < n1; i++) d1[k][i] = coef[k] * sgn(p1[k][i] - g1[k][i]);
  }
  for (size_t i = 0; i < n1; i++) d1[0][i] += dll1[i];
  dwt_level_adj(d1[0].data(), d1[1].data(), d1[2].data(), d1[3].data(), C, H, W, grad_pred, accumulate != 0);
  return GS_OK;
}

int gso_elf_map(const float* img, int32_t C, int32_t H, int32_t W, float* elf_low, float* elf, void*) {
  if (!img || !elf) return GS_E_NULL;
  const int h = cdiv2(H), w = cdiv2(W);
  const size_t n1 = (size_t)C * h * w;
  std::vector<float> b[4];
  for (int k = 0; k < 4; k++) b[k].resize(n1);
  dwt_level(img, C, H, W, b[0].data(), b[1].data(), b[2].data(), b[3].data());
  std::vector<float> low((size_t)h * w);
  for (int i = 0; i < h * w; i++) {
    float s[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++)
      for (int c = 0; c < C; c++) s[k] += fabsf(b[k][(size_t)c * h * w + i]);
    float HF = s[1] + s[2] + s[3];
    low[i] = s[0] / (s[0] + HF + 1e-8f);
  }
  if (elf_low) memcpy(elf_low, low.data(), sizeof(float) * h * w);
  /* F.interpolate(size=(H,W), mode='bilinear', align_corners=False): ATen upsample_bilinear2d */
  const float sh = (float)h / (float)H, sw = (float)w / (float)W;
  for (int y = 0; y < H; y++) {
    float sy = sh * ((float)y + 0.5f) - 0.5f;
    if (sy < 0) sy = 0;
    int y0 = (int)sy, y1 = y0 + (y0 < h - 1 ? 1 : 0);
    float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
    for (int x = 0; x < W; x++) {
      float sx = sw * ((float)x + 0.5f) - 0.5f;
      if (sx < 0) sx = 0;
      int x0 = (int)sx, x1 = x0 + (x0 < w - 1 ? 1 : 0);
      float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
      elf[(size_t)y * W + x] = ly0 * (lx0 * low[(size_t)y0 * w + x0] + lx1 * low[(size_t)y0 * w + x1]) +
                               ly1 * (lx0 * low[(size_t)y1 * w + x0] + lx1 * low[(size_t)y1 * w + x1]);
    }
  }
  return GS_OK;
}

int gso_patch_means(const float* elf, int32_t H, int32_t W, int32_t ps, float* means, void*) {
  if (!elf || !means) return GS_E_NULL;
  const int ny = H / ps, nx = W / ps;
  for (int py = 0; py < ny; py++)
    for (int px = 0; px < nx; px++) {
      double s = 0;
      for (int y = 0; y < ps; y++)
        for (int x = 0; x < ps; x++) s += (double)elf[(size_t)(py * ps + y) * W + px * ps + x];
      means[py * nx + px] = (float)(s / ((double)ps * ps));
    }
  return GS_OK;
}

static void patch_bands(const float* img, int C, int H, int W, int ps, int py, int px, int c, int i, int j, Bands& b) {
  const float* p = img + (size_t)c * H * W;
  int y0 = py * ps + 2 * i, x0 = px * ps + 2 * j;
  /* patch size is even (128) in every reference configuration; odd patches pad inside the patch */
  int y1 = (2 * i + 1 < ps) ? y0 + 1 : y0, x1 = (2 * j + 1 < ps) ? x0 + 1 : x0;
  b = haar_block(p[(size_t)y0 * W + x0], p[(size_t)y0 * W + x1], p[(size_t)y1 * W + x0], p[(size_t)y1 * W + x1]);
  (void)C;
}

int gso_patch_dwt_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps,
                      const uint8_t* mask, float* sums, void*) {
  if (!pred || !gt || !mask || !sums) return GS_E_NULL;
  const int ny = H / ps, nx = W / ps, hp = cdiv2(ps);
  double s[3] = {0, 0, 0};
  for (int py = 0; py < ny; py++)
    for (int px = 0; px < nx; px++) {
      if (!mask[py * nx + px]) continue;
      for (int c = 0; c < C; c++)
        for (int i = 0; i < hp; i++)
          for (int j = 0; j < hp; j++) {
            Bands a, b;
            patch_bands(pred, C, H, W, ps, py, px, c, i, j, a);
            patch_bands(gt, C, H, W, ps, py, px, c, i, j, b);
            s[0] += (double)fabsf(a.lh - b.lh);
            s[1] += (double)fabsf(a.hl - b.hl);
            s[2] += (double)fabsf(a.hh - b.hh);
          }
    }
  for (int k = 0; k < 3; k++) sums[k] += (float)s[k];
  return GS_OK;
}

int gso_patch_dwt_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps,
                      const uint8_t* mask, const float* coef, float* grad, int32_t accumulate, void*) {
  if (!pred || !gt || !mask || !coef || !grad) return GS_E_NULL;
  const int ny = H / ps, nx = W / ps, hp = cdiv2(ps);
  if (!accumulate) memset(grad, 0, sizeof(float) * (size_t)C * H * W);
  for (int py = 0; py < ny; py++)
    for (int px = 0; px < nx; px++) {
      if (!mask[py * nx + px]) continue;
      for (int c = 0; c < C; c++)
        for (int i = 0; i < hp; i++)
          for (int j = 0; j < hp; j++) {
            Bands a, b;
            patch_bands(pred, C, H, W, ps, py, px, c, i, j, a);
            patch_bands(gt, C, H, W, ps, py, px, c, i, j, b);
            float da, db, dc, dd;
            haar_block_adj(0.f, coef[0] * sgn(a.lh - b.lh), coef[1] * sgn(a.hl - b.hl), coef[2] * sgn(a.hh - b.hh),
                           da, db, dc, dd);
            float* p = grad + (size_t)c * H * W;
            int y0 = py * ps + 2 * i, x0 = px * ps + 2 * j;
            int y1 = (2 * i + 1 < ps) ? y0 + 1 : y0, x1 = (2 * j + 1 < ps) ? x0 + 1 : x0;
            p[(size_t)y0 * W + x0] += da;
            p[(size_t)y0 * W + x1] += db;
            p[(size_t)y1 * W + x0] += dc;
            p[(size_t)y1 * W + x1] += dd;
          }
    }
  return GS_OK;
}

/* fused-ssim/ssim.cu:187-286 */
int gso_ssim_fwd(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float C1,
                 float C2, float* ssim_map, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, void*) {
  if (!img1 || !img2 || !ssim_map) return GS_E_NULL;
  const size_t np = (size_t)H * W;
  std::vector<float> tmp, a(np), mu1(np), mu2(np), s11(np), s22(np), s12(np);
  for (int bc = 0; bc < B * C; bc++) {
    const float* x = img1 + bc * np;
    const float* y = img2 + bc * np;
    sepconv(x, H, W, tmp, mu1.data());
    for (size_t i = 0; i < np; i++) a[i] = x[i] * x[i];
    sepconv(a.data(), H, W, tmp, s11.data());
    sepconv(y, H, W, tmp, mu2.data());
    for (size_t i = 0; i < np; i++) a[i] = y[i] * y[i];
    sepconv(a.data(), H, W, tmp, s22.data());
    for (size_t i = 0; i < np; i++) a[i] = x[i] * y[i];
    sepconv(a.data(), H, W, tmp, s12.data());
    for (size_t i = 0; i < np; i++) {
      float m1 = mu1[i], m2 = mu2[i];
      float sigma1_sq = s11[i] - m1 * m1;
      float sigma2_sq = s22[i] - m2 * m2;
      float sigma12 = s12[i] - m1 * m2;
      float mu1_sq = m1 * m1, mu2_sq = m2 * m2, mu1_mu2 = m1 * m2;
      float Cc = (2.0f * mu1_mu2 + C1);
      float D = (2.0f * sigma12 + C2);
      float A = (mu1_sq + mu2_sq + C1);
      float Bb = (sigma1_sq + sigma2_sq + C2);
      float m = (Cc * D) / (A * Bb);
      size_t o = bc * np + i;
      ssim_map[o] = m;
      if (dm_dmu1) {
        dm_dmu1[o] = ((m2 * 2.0f * D) / (A * Bb) - (m2 * 2.0f * Cc) / (A * Bb) - (m1 * 2.0f * Cc * D) / (A * A * Bb) +
                      (m1 * 2.0f * Cc * D) / (A * Bb * Bb));
        dm_dsigma1_sq[o] = ((-Cc * D) / (A * Bb * Bb));
        dm_dsigma12[o] = ((2 * Cc) / (A * Bb));
      }
    }
  }
  return GS_OK;
}

/* fused-ssim/ssim.cu:288-366 */
int gso_ssim_bwd(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float, float,
                 const float* dL_dmap, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12,
                 float* dL_dimg1, void*) {
  if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) return GS_E_NULL;
  const size_t np = (size_t)H * W;
  std::vector<float> tmp, a(np), r(np);
  for (int bc = 0; bc < B * C; bc++) {
    const size_t o = bc * np;
    for (size_t i = 0; i < np; i++) a[i] = dm_dmu1[o + i] * dL_dmap[o + i];
    sepconv(a.data(), H, W, tmp, r.data());
    for (size_t i = 0; i < np; i++) dL_dimg1[o + i] = 0.0f + r[i];
    for (size_t i = 0; i < np; i++) a[i] = dm_dsigma1_sq[o + i] * dL_dmap[o + i];
    sepconv(a.data(), H, W, tmp, r.data());
    for (size_t i = 0; i < np; i++) dL_dimg1[o + i] += img1[o + i] * 2.0f * r[i];
    for (size_t i = 0; i < np; i++) a[i] = dm_dsigma12[o + i] * dL_dmap[o + i];
    sepconv(a.data(), H, W, tmp, r.data());
    for (size_t i = 0; i < np; i++) dL_dimg1[o + i] += img2[o + i] * r[i];
  }
  return GS_OK;
}

/* torch.optim.Adam arithmetic (no amsgrad / weight decay) over a flat buffer with a learning-rate segment table;
 * restates the update of LGDWT-GS/scene/gaussian_model.py:183-193 + train.py:279-288 (pinned against
 * torch.optim.Adam itself in tests/test_adam.py) */
int gso_adam_step_masked(float* p, const float* g, float* m, float* v, int64_t n, const GsAdamSeg* segs, int32_t nseg,
                         float b1, float b2, float eps, int32_t step, const float* gate, const float* row_mask, void*);
int gso_adam_step(float* p, const float* g, float* m, float* v, int64_t n, const GsAdamSeg* segs, int32_t nseg,
                  float b1, float b2, float eps, int32_t step, void*) {
  return gso_adam_step_masked(p, g, m, v, n, segs, nseg, b1, b2, eps, step, nullptr, nullptr, nullptr);
}
/* + "sparse_adam" (LGDWT-GS/train.py:282-284: visible = radii > 0; optimizer.step(visible, N)): rows with row_mask <= 0 keep
 * parameters and moments; gate != 0: nobody steps (data-parallel validity flag) */
int gso_adam_step_masked(float* p, const float* g, float* m, float* v, int64_t n, const GsAdamSeg* segs, int32_t nseg,
                         float b1, float b2, float eps, int32_t step, const float* gate, const float* row_mask, void*) {
  if (!p || !g || !m || !v) return GS_E_NULL;
  if (gate && *gate != 0.0f) return GS_OK;
  if (step < 1 || nseg < 0 || nseg > 8) return GS_E_SHAPE;
  float seg_inv_bc1[8], seg_inv_sqrt_bc2[8];
  for (int k = 0; k < nseg; k++) {
    const int st = segs[k].step > 0 ? segs[k].step : step;  // torch keeps the step count per parameter
    const double bc1 = 1.0 - pow((double)b1, (double)st), bc2 = 1.0 - pow((double)b2, (double)st);
    seg_inv_bc1[k] = (float)(1.0 / bc1);
    seg_inv_sqrt_bc2[k] = (float)(1.0 / sqrt(bc2));
  }
  for (int64_t i = 0; i < n; i++) {
    float lr = 0.f;
    int seg = -1;
    for (int k = 0; k < nseg; k++)
      if (i >= segs[k].begin && i < segs[k].end) {
        lr = (segs[k].period > 0 && (int)((i - segs[k].begin) % segs[k].period) >= segs[k].split) ? segs[k].lr_b : segs[k].lr_a;
        seg = k;
      }
    if (seg < 0) continue;  // not optimised this step (torch skips parameters without a gradient)
    if (row_mask && segs[seg].row_width > 0 && !(row_mask[(i - segs[seg].begin) / segs[seg].row_width] > 0.f)) continue;
    const float inv_bc1 = seg_inv_bc1[seg], inv_sqrt_bc2 = seg_inv_sqrt_bc2[seg];
    m[i] = b1 * m[i] + (1.f - b1) * g[i];
    v[i] = b2 * v[i] + (1.f - b2) * g[i] * g[i];
    const float denom = sqrtf(v[i]) * inv_sqrt_bc2 + eps;
    p[i] = p[i] - (lr * inv_bc1) * (m[i] / denom);
  }
  return GS_OK;
}

/* the visibility-sparse gradient exchange's pack / unpack (include/gsplat.h; product-side feature, mirrored so that the CPU
 * data-parallel tests run the same glue) */
static int rows_pack_cpu(float* flat, int32_t P, int32_t nf, const int32_t* widths, const uint8_t* mask, const int32_t* pos,
                         int32_t K, float* packed, bool unpack) {
  if (P < 0 || K < 0 || nf < 0 || nf > 8) return GS_E_SHAPE;
  if (P == 0 || K == 0 || nf == 0) return GS_OK;
  if (!flat || !widths || !mask || !pos || !packed) return GS_E_NULL;
  int64_t fo = 0, po = 0;
  for (int q = 0; q < nf; q++) {
    const int w = widths[q];
    for (int i = 0; i < P; i++) {
      if (!mask[i] || pos[i] < 0 || pos[i] >= K) continue;
      float* a = flat + fo + (int64_t)i * w;
      float* b = packed + po + (int64_t)pos[i] * w;
      for (int j = 0; j < w; j++) {
        if (unpack) a[j] = b[j]; else b[j] = a[j];
      }
    }
    fo += (int64_t)P * w;
    po += (int64_t)K * w;
  }
  return GS_OK;
}
int gso_rows_pack(const float* flat, int32_t P, int32_t nf, const int32_t* widths, const uint8_t* mask, const int32_t* pos,
                  int32_t K, float* packed, void*) {
  return rows_pack_cpu(const_cast<float*>(flat), P, nf, widths, mask, pos, K, packed, false);
}
int gso_rows_unpack(float* flat, int32_t P, int32_t nf, const int32_t* widths, const uint8_t* mask, const int32_t* pos, int32_t K,
                    const float* packed, void*) {
  return rows_pack_cpu(flat, P, nf, widths, mask, pos, K, const_cast<float*>(packed), true);
}

/* ---- per-Gaussian elementwise work of the train step (restates LGDWT-GS/scene/gaussian_model.py:40-60,102-117
 * activations with torch's autograd formulas, and train.py:266-268 + gaussian_model.py:471-473 statistics;
 * pinned against torch itself in tests/test_model_ops.py) ---- */
int gso_activations_fwd(const float* scaling, const float* rotation, const float* opacity, int32_t P, float* o_scales,
                        float* o_rot, float* o_opac, void*) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!scaling || !rotation || !opacity || !o_scales || !o_rot || !o_opac) return GS_E_NULL;
  for (int i = 0; i < P; i++) {
    for (int k = 0; k < 3; k++) o_scales[3 * i + k] = expf(scaling[3 * i + k]);
    const float* q = rotation + 4 * (size_t)i;
    const float n = fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
    for (int k = 0; k < 4; k++) o_rot[4 * (size_t)i + k] = q[k] / n;
    o_opac[i] = 1.0f / (1.0f + expf(-opacity[i]));
  }
  return GS_OK;
}
int gso_activations_bwd(const float* scaling, const float* rotation, const float* opacity, int32_t P, const float* g_scales,
                        const float* g_rot, const float* g_opac, float* d_scaling, float* d_rotation, float* d_opacity,
                        void*) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!scaling || !rotation || !opacity || !g_scales || !g_rot || !g_opac || !d_scaling || !d_rotation || !d_opacity)
    return GS_E_NULL;
  for (int i = 0; i < P; i++) {
    for (int k = 0; k < 3; k++) d_scaling[3 * i + k] = g_scales[3 * i + k] * expf(scaling[3 * i + k]);
    const float* q = rotation + 4 * (size_t)i;
    const float* g = g_rot + 4 * (size_t)i;
    float* d = d_rotation + 4 * (size_t)i;
    const float norm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (norm > 1e-12f) {
      const float inv = 1.0f / norm;
      const float v[4] = {q[0] * inv, q[1] * inv, q[2] * inv, q[3] * inv};
      const float dot = v[0] * g[0] + v[1] * g[1] + v[2] * g[2] + v[3] * g[3];
      for (int k = 0; k < 4; k++) d[k] = (g[k] - v[k] * dot) * inv;
    } else {
      for (int k = 0; k < 4; k++) d[k] = g[k] / 1e-12f;
    }
    const float s = 1.0f / (1.0f + expf(-opacity[i]));
    d_opacity[i] = g_opac[i] * (1.0f - s) * s;
  }
  return GS_OK;
}
int gso_densify_stats(const int32_t* radii, const float* dL_dmeans2D, int32_t P, float* max_radii2D, float* accum,
                      float* denom, void*) {
  if (P < 0) return GS_E_SHAPE;
  if (P == 0) return GS_OK;
  if (!radii || !dL_dmeans2D || !max_radii2D || !accum || !denom) return GS_E_NULL;
  for (int i = 0; i < P; i++)
    if (radii[i] > 0) {
      max_radii2D[i] = fmaxf(max_radii2D[i], (float)radii[i]);
      const float gx = dL_dmeans2D[3 * i], gy = dL_dmeans2D[3 * i + 1];
      accum[i] += sqrtf(gx * gx + gy * gy);
      denom[i] += 1.0f;
    }
  return GS_OK;
}

/* ---- fused-criterion entry points (same arithmetic as the per-term functions above) ---- */
int gso_l1_bwd_dev(const float* a, const float* b, int64_t n, const float* coef, float* g, int32_t accumulate, void*) {
  if (!a || !b || !g || !coef) return GS_E_NULL;
  return gso_l1_bwd(a, b, n, 1.0f * coef[0], g, accumulate, nullptr);
}
int gso_l1_dwt2_fwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, float* l1_sum, float* band_sums,
                    void*) {
  if (!l1_sum) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  int rc = gso_l1_fwd(pred, gt, (int64_t)C * H * W, l1_sum, nullptr);
  return rc ? rc : gso_dwt2_l1_fwd(pred, gt, C, H, W, band_sums, nullptr);
}
int gso_l1_dwt2_fwd_clamp(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, float* l1_sum, float* band_sums,
                          float* clamped_out, void*) {
  if (!raw || !clamped_out) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  if ((H % 4) != 0 || (W % 4) != 0) return GS_E_UNSUPPORTED;
  const int64_t n = (int64_t)C * H * W;
  for (int64_t i = 0; i < n; i++) clamped_out[i] = fminf(fmaxf(raw[i], 0.f), 1.f);  // gaussian_renderer/__init__.py:119
  return gso_l1_dwt2_fwd(clamped_out, gt, C, H, W, l1_sum, band_sums, nullptr);
}
int gso_l1_dwt2_patch_fwd_clamp(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps, const uint8_t* mask,
                                float* l1_sum, float* band_sums, float* patch_sums, float* clamped_out, void*) {
  if (!mask || !patch_sums) return GS_E_NULL;
  if (ps <= 0 || (ps % 4) != 0) return ps <= 0 ? GS_E_SHAPE : GS_E_UNSUPPORTED;
  int rc = gso_l1_dwt2_fwd_clamp(raw, gt, C, H, W, l1_sum, band_sums, clamped_out, nullptr);
  return rc ? rc : gso_patch_dwt_fwd(clamped_out, gt, C, H, W, ps, mask, patch_sums, nullptr);
}
int gso_l1_dwt2_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, const float* l1_coef,
                    const float* coef, float* grad, int32_t accumulate, void*) {
  if (!l1_coef) return GS_E_NULL;
  if (C <= 0 || H <= 0 || W <= 0) return GS_E_SHAPE;
  int rc = gso_l1_bwd_dev(pred, gt, (int64_t)C * H * W, l1_coef, grad, accumulate, nullptr);
  return rc ? rc : gso_dwt2_l1_bwd(pred, gt, C, H, W, coef, grad, 1, nullptr);
}
int gso_l1_dwt2_patch_bwd(const float* pred, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps, const uint8_t* mask,
                          const float* l1_coef, const float* coef, const float* patch_coef, float* grad, int32_t accumulate, void*) {
  if (!mask || !patch_coef) return GS_E_NULL;
  if (ps <= 0 || (ps % 4) != 0) return ps <= 0 ? GS_E_SHAPE : GS_E_UNSUPPORTED;
  if ((H % 4) != 0 || (W % 4) != 0) return GS_E_UNSUPPORTED;
  int rc = gso_l1_dwt2_bwd(pred, gt, C, H, W, l1_coef, coef, grad, accumulate, nullptr);
  return rc ? rc : gso_patch_dwt_bwd(pred, gt, C, H, W, ps, mask, patch_coef, grad, 1, nullptr);
}
int gso_ssim_fwd_sum(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float C1, float C2,
                     float* sum_out, float* d1, float* d2, float* d3, void*) {
  if (!sum_out) return GS_E_NULL;
  std::vector<float> map((size_t)B * C * H * W);
  int rc = gso_ssim_fwd(img1, img2, B, C, H, W, C1, C2, map.data(), d1, d2, d3, nullptr);
  if (rc) return rc;
  double s = 0;
  for (float v : map) s += (double)v;
  *sum_out += (float)s;
  return GS_OK;
}
int gso_ssim_bwd_uniform(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, const float* coef,
                         const float* d1, const float* d2, const float* d3, float* dL_dimg1, int32_t accumulate,
                         const float* clamp_src, void*) {
  if (!coef || !dL_dimg1) return GS_E_NULL;
  const size_t n = (size_t)B * C * H * W;
  std::vector<float> g(n, coef[0]), out(n);
  int rc = gso_ssim_bwd(img1, img2, B, C, H, W, 0.f, 0.f, g.data(), d1, d2, d3, out.data(), nullptr);
  if (rc) return rc;
  for (size_t i = 0; i < n; i++) {
    float v = out[i];
    if (accumulate) v += dL_dimg1[i];
    if (clamp_src && (clamp_src[i] < 0.f || clamp_src[i] > 1.f)) v = 0.f;
    dL_dimg1[i] = v;
  }
  return GS_OK;
}
/* per-workgroup partial sums of the HIP kernel: the checker puts the whole sum into partials[0] */
int64_t gso_ssim_partials_count(int32_t B, int32_t C, int32_t H, int32_t W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  return (int64_t)((W + 31) / 32) * ((H + 31) / 32) * B * C;
}
int gso_ssim_fwd_partials(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W, float C1, float C2,
                          float* partials, float* d1, float* d2, float* d3, void*) {
  if (!partials) return GS_E_NULL;
  const int64_t n = gso_ssim_partials_count(B, C, H, W);
  for (int64_t i = 0; i < n; i++) partials[i] = 0.f;
  return gso_ssim_fwd_sum(img1, img2, B, C, H, W, C1, C2, partials, d1, d2, d3, nullptr);
}
/* LGDWT-GS/train.py:188-202 */
int gso_lgdwt_combine(const float* sums, float* running_mean, const GsLgdwtParams* pp, float* out, void*) {
  if (!sums || !running_mean || !pp || !out) return GS_E_NULL;
  const GsLgdwtParams& p = *pp;
  const float l1 = sums[0] / p.n_pix;
  const float ssim = sums[1] / p.n_pix;
  /* custom_base: the NIR term of mult-dwtgs/train_nir.py:96-104 (w_l1 L1 + w_ssim (1 - SSIM)) */
  const float w_l1 = p.custom_base ? p.w_l1 : 1.0f - p.lambda_dssim, w_ssim = p.custom_base ? p.w_ssim : p.lambda_dssim;
  const float base = w_l1 * l1 + w_ssim * (1.0f - ssim);
  float loss = base, dwt = 0.f, scale = 0.f, patch = 0.f;
  for (int k = 0; k < 24; k++) out[k] = 0.f;
  out[7] = running_mean[0];
  if (p.dwt_enable) {
    for (int k = 0; k < 8; k++) dwt += p.dwt_w[k] * (sums[2 + k] / (k < 4 ? p.n_band1 : p.n_band2));
    const float ratio = base / (dwt + 1e-8f);
    const float m = 0.95f * running_mean[0] + 0.05f * ratio;
    running_mean[0] = m;
    scale = fminf(fmaxf(m, 0.1f), 10.0f);
    loss = base + scale * dwt;
    for (int k = 0; k < 8; k++) out[10 + k] = scale * p.dwt_w[k] / (k < 4 ? p.n_band1 : p.n_band2);
  }
  if (p.patch_enable) {
    const float denom = fmaxf(sums[13] * p.patch_elems_per_sel, 1.0f);
    for (int k = 0; k < 3; k++) {
      patch += p.patch_w[k] * (sums[10 + k] / denom);
      out[18 + k] = p.patch_weight * p.patch_w[k] / denom;
    }
    loss = loss + p.patch_weight * patch;
  }
  out[0] = loss; out[1] = base; out[2] = dwt; out[3] = patch; out[4] = scale; out[5] = l1; out[6] = ssim;
  out[8] = w_l1 / p.n_pix;
  out[9] = -w_ssim / p.n_pix;
  if (p.reset_sums)
    for (int k = 0; k < 13; k++) const_cast<float*>(sums)[k] = 0.f;
  return GS_OK;
}
int gso_lgdwt_combine_p(const float* sums, const float* ssim_partials, int64_t n_partials, float* running_mean,
                        const GsLgdwtParams* pp, float* out, void*) {
  if (!sums) return GS_E_NULL;
  if (n_partials < 0 || (n_partials > 0 && !ssim_partials)) return GS_E_SHAPE;
  float s2[16];
  for (int k = 0; k < 16; k++) s2[k] = sums[k];
  double add = 0;
  for (int64_t i = 0; i < n_partials; i++) add += (double)ssim_partials[i];
  s2[1] += (float)add;
  const int rc = gso_lgdwt_combine(s2, running_mean, pp, out, nullptr);
  if (rc == GS_OK && pp->reset_sums)
    for (int k = 0; k < 13; k++) const_cast<float*>(sums)[k] = 0.f;
  return rc;
}
/* The order-independent forms (include/gsplat.h): on the CPU the sums are sequential anyway - ONE row of partials. */
int64_t gso_dwt_partials_count(int32_t C, int32_t H, int32_t W) { return (C <= 0 || H <= 0 || W <= 0) ? 0 : 1; }
int gso_l1_dwt2_patch_fwd_clamp_p(const float* raw, const float* gt, int32_t C, int32_t H, int32_t W, int32_t ps,
                                  const uint8_t* mask, float* partials, float* clamped_out, void*) {
  if (!raw || !gt || !partials || !clamped_out) return GS_E_NULL;
  if ((ps > 0) != (mask != nullptr)) return GS_E_NULL;
  float l1 = 0.f, bands[8] = {0, 0, 0, 0, 0, 0, 0, 0}, patch[3] = {0, 0, 0};
  int rc;
  if (ps > 0)
    rc = gso_l1_dwt2_patch_fwd_clamp(raw, gt, C, H, W, ps, mask, &l1, bands, patch, clamped_out, nullptr);
  else
    rc = gso_l1_dwt2_fwd_clamp(raw, gt, C, H, W, &l1, bands, clamped_out, nullptr);
  if (rc != GS_OK) return rc;
  for (int k = 0; k < 8; k++) partials[k] = bands[k];
  partials[8] = l1;
  for (int k = 0; k < 3; k++) partials[9 + k] = patch[k];
  return GS_OK;
}
int64_t gso_l1_partials_count(int64_t n) { return n <= 0 ? 0 : 1; }
/* Depth regularisation, LGDWT-GS/train.py:204-216: Ll1depth_pure = torch.abs((invDepth - mono_invdepth) * depth_mask).mean()
 * -> partials[0] = the sum (one "workgroup"); grad = coef * sign((d - m) k) k (torch.abs backward: sign(0) = 0) */
int64_t gso_depth_l1_partials_count(int64_t n) { return n <= 0 ? 0 : 1; }
int gso_depth_l1(const float* d, const float* m, const float* k, int64_t n, float* partials, float coef, const float* coef_dev,
                 float* grad, void*) {
  if (!d || !m || (!partials && !grad)) return GS_E_NULL;
  if (n <= 0) return GS_OK;
  if (coef_dev) coef *= coef_dev[0];
  double s = 0.0;
  for (int64_t i = 0; i < n; i++) {
    const float w = k ? k[i] : 1.0f;
    const float v = (d[i] - m[i]) * w;
    s += (double)fabsf(v);
    if (grad) grad[i] = coef * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f)) * w;
  }
  if (partials) partials[0] = (float)s;
  return GS_OK;
}
int gso_l1_fwd_p(const float* a, const float* b, int64_t n, float* partials, void*) {
  if (!a || !b || !partials) return GS_E_NULL;
  partials[0] = 0.f;
  return gso_l1_fwd(a, b, n, partials, nullptr);
}
int gso_lgdwt_combine_pp(const float* sums, const float* ssim_partials, int64_t n_partials, const float* dwt_partials,
                         int64_t n_dwt, const float* l1_partials, int64_t n_l1, float* running_mean, const GsLgdwtParams* pp,
                         float* out, void*) {
  if (!sums) return GS_E_NULL;
  float s2[16];
  for (int k = 0; k < 16; k++) s2[k] = sums[k];
  for (int64_t w = 0; w < n_dwt; w++) {
    for (int k = 0; k < 8; k++) s2[2 + k] += dwt_partials[12 * w + k];
    s2[0] += dwt_partials[12 * w + 8];
    for (int k = 0; k < 3; k++) s2[10 + k] += dwt_partials[12 * w + 9 + k];
  }
  for (int64_t w = 0; w < n_l1; w++) s2[0] += l1_partials[w];
  GsLgdwtParams p2 = *pp;
  p2.reset_sums = 0;
  const int rc = gso_lgdwt_combine_p(s2, ssim_partials, n_partials, running_mean, &p2, out, nullptr);
  if (rc == GS_OK && pp->reset_sums)
    for (int k = 0; k < 13; k++) const_cast<float*>(sums)[k] = 0.f;
  return rc;
}
}
