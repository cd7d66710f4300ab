// Hoeffding-Bentkus upper confidence bound on a [0,1] mean, host float64.
//
// Replaces HB_mu_plus (core/calibration/bounds.py:17-29) and its helpers h1 (:6-7),
// hoeffding_plus (:10-11), bentkus_plus (:13-14).  The reference reaches scipy for the two
// non-trivial pieces; both are restated here from their published algorithms:
//   * binom.cdf(k, n, p) = I_{1-p}(n-k, k+1), the regularised incomplete beta function,
//     evaluated by the modified-Lentz continued fraction (DLMF 8.17.22) with an lgamma prefactor;
//   * scipy.optimize.brentq = Brent's 1973 bracketing root finder (inverse quadratic / secant /
//     bisection) with scipy's default xtol = 2e-12, rtol = 4*eps.
// Python-level semantics that matter are kept: builtin min/max NaN behaviour, and "any solver
// exception -> 1.0" (NaN function value, same-sign bracket, no convergence).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <thread>
#include <vector>
#include "../../include/im2im_uq.h"

namespace {

// continued fraction for I_x(a,b), valid/fast for x < (a+1)/(a+b+2)
double betacf(double a, double b, double x) {
  const double tiny = 1e-300, eps = 1e-16;
  double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (std::fabs(d) < tiny) d = tiny;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 100000; ++m) {
    double m2 = 2.0 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d; if (std::fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d; h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d; if (std::fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (std::fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (std::fabs(del - 1.0) < eps) break;
  }
  return h;
}

// regularised incomplete beta I_x(a,b), a,b > 0, 0 <= x <= 1
double ibeta(double a, double b, double x) {
  if (x <= 0.0) return 0.0;
  if (x >= 1.0) return 1.0;
  double lbt = std::lgamma(a + b) - std::lgamma(a) - std::lgamma(b) + a * std::log(x) + b * std::log1p(-x);
  double bt = std::exp(lbt);
  if (x < (a + 1.0) / (a + b + 2.0)) return bt * betacf(a, b, x) / a;
  return 1.0 - bt * betacf(b, a, 1.0 - x) / b;
}

// scipy.stats.binom.cdf(k, n, p) for real k (already floored), integer n
double binom_cdf(double k, double n, double p) {
  if (std::isnan(k) || std::isnan(p)) return std::numeric_limits<double>::quiet_NaN();
  if (k < 0) return 0.0;
  if (k >= n) return 1.0;
  if (p <= 0.0) return 1.0;
  if (p >= 1.0) return 0.0;
  return ibeta(n - k, k + 1.0, 1.0 - p);
}

inline double py_min(double a, double b) { return (b < a) ? b : a; }  // builtin min(a, b)
inline double py_max(double a, double b) { return (b > a) ? b : a; }  // builtin max(a, b)

struct Tail {
  double muhat, n, log_delta;
  double k = 0.0;            // floor(n * muhat) supplied by the caller (batch form: the reference's fp32 product)
  bool has_k = false;
  // _tailprob, bounds.py:18-21
  double operator()(double mu) const {
    double y = std::fmin(mu, muhat);                       // np.minimum(mu, x)  (bounds.py:11)
    if (std::isnan(mu) || std::isnan(muhat)) y = std::numeric_limits<double>::quiet_NaN();
    double h1 = y * std::log(y / mu) + (1.0 - y) * std::log((1.0 - y) / (1.0 - mu));   // bounds.py:7
    double hoeffding = -n * h1;
    double bentkus = std::log(py_max(binom_cdf(has_k ? k : std::floor(n * muhat), n, mu), 1e-10)) + 1.0;  // bounds.py:14
    return py_min(hoeffding, bentkus) - log_delta;
  }
};

// Brent's method as scipy.optimize.brentq runs it; ok=false stands for "scipy raised".
double brentq(const Tail& f, double xa, double xb, double xtol, double rtol, int maxiter, bool& ok) {
  ok = false;
  double xpre = xa, xcur = xb, xblk = 0.0, fblk = 0.0, spre = 0.0, scur = 0.0;
  double fpre = f(xpre), fcur = f(xcur);
  if (std::isnan(fpre) || std::isnan(fcur)) return 0.0;     // "function value is NaN; solver cannot continue"
  if (fpre == 0.0) { ok = true; return xpre; }
  if (fcur == 0.0) { ok = true; return xcur; }
  if (std::signbit(fpre) == std::signbit(fcur)) return 0.0;  // "f(a) and f(b) must have different signs"
  for (int i = 0; i < maxiter; ++i) {
    if (fpre != 0.0 && fcur != 0.0 && std::signbit(fpre) != std::signbit(fcur)) {
      xblk = xpre; fblk = fpre; spre = scur = xcur - xpre;
    }
    if (std::fabs(fblk) < std::fabs(fcur)) {
      xpre = xcur; xcur = xblk; xblk = xpre;
      fpre = fcur; fcur = fblk; fblk = fpre;
    }
    double delta = (xtol + rtol * std::fabs(xcur)) / 2.0;
    double sbis = (xblk - xcur) / 2.0;
    if (fcur == 0.0 || std::fabs(sbis) < delta) { ok = true; return xcur; }
    if (std::fabs(spre) > delta && std::fabs(fcur) < std::fabs(fpre)) {
      double stry;
      if (xpre == xblk) {
        stry = -fcur * (xcur - xpre) / (fcur - fpre);                       // secant
      } else {
        double dpre = (fpre - fcur) / (xpre - xcur);                        // inverse quadratic
        double dblk = (fblk - fcur) / (xblk - xcur);
        stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
      }
      if (2.0 * std::fabs(stry) < std::fmin(std::fabs(spre), 3.0 * std::fabs(sbis) - delta)) {
        spre = scur; scur = stry;
      } else {
        spre = sbis; scur = sbis;
      }
    } else {
      spre = sbis; scur = sbis;
    }
    xpre = xcur; fpre = fcur;
    if (std::fabs(scur) > delta) xcur += scur;
    else xcur += (sbis > 0 ? delta : -delta);
    fcur = f(xcur);
    if (std::isnan(fcur)) return 0.0;
  }
  return 0.0;  // "Failed to converge"
}

}  // namespace

extern "C" double im2im_hb_mu_plus(double muhat, int64_t n, double delta, int32_t maxiters) {
  Tail f{muhat, (double)n, std::log(delta)};
  const double hi = 1.0 - 1e-10;
  double fhi = f(hi);
  if (fhi > 0.0) return 1.0;                                 // bounds.py:22-23 (NaN > 0 is False)
  bool ok = false;
  double root = brentq(f, muhat, hi, 2e-12, 4.0 * std::numeric_limits<double>::epsilon(), maxiters, ok);
  return ok ? root : 1.0;                                    // bounds.py:25-29
}

// A whole row of bounds at once (evaluate_from_loss_table, core/calibration/calibrate_model.py:62-74, solves one per lambda,
// and experiments/fastmri_test/plot.py:126-139 repeats that 100 times): muhat[i] are float32 empirical risks as the reference
// holds them (0-dim fp32 tensors), so floor(n * muhat) is taken on the fp32 product like `np.floor(n * muhat)` there; the
// solves are independent and are spread over host threads.
extern "C" int im2im_hb_mu_plus_batch(const float* muhat, int64_t count, int64_t n, double delta, int32_t maxiters,
                                      double* out) {
  if (!muhat || !out || count < 0 || n <= 0) return -1;
  auto solve = [&](int64_t i) {
    const float m32 = muhat[i];
    Tail f{(double)m32, (double)n, std::log(delta)};
    f.k = std::floor((double)((float)n * m32));
    f.has_k = true;
    const double hi = 1.0 - 1e-10;
    if (f(hi) > 0.0) { out[i] = 1.0; return; }
    bool ok = false;
    const double root = brentq(f, (double)m32, hi, 2e-12, 4.0 * std::numeric_limits<double>::epsilon(), maxiters, ok);
    out[i] = ok ? root : 1.0;
  };
  unsigned hw = std::thread::hardware_concurrency();
  const int64_t nthreads = std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), (int64_t)16, count / 8}));
  if (nthreads == 1) {
    for (int64_t i = 0; i < count; ++i) solve(i);
    return 0;
  }
  std::vector<std::thread> pool;
  for (int64_t t = 0; t < nthreads; ++t)
    pool.emplace_back([&, t]() { for (int64_t i = t; i < count; i += nthreads) solve(i); });
  for (auto& th : pool) th.join();
  return 0;
}

// The descending lambda scan of calibrate_model (core/calibration/calibrate_model.py:130-144) over a finished loss table, so
// that a non-Python caller of this library obtains lambda-hat too.  Reference loop, quirks kept:
//     lhat = lambdas[-1] + dlambda - 1e-9                              (default when nothing stops the scan; fp32 arithmetic,
//                                                                       so the 1e-9 vanishes: Q4)
//     for lam in reversed(lambdas):
//         losses = losses_at(lam - dlambda)                            (the shifted grid, Q1: column j of `table` holds it)
//         Rhat = losses.mean(); RhatPlus = HB_mu_plus(Rhat.item(), n, delta)      (Rhat == 0 -> RhatPlus = 1.0, Q3)
//         if Rhat >= alpha or RhatPlus > alpha: lhat = lam; break      (fp32 Rhat against fp32(alpha); the bound in float64)
// table(i, j) = table[i * row_stride + j * col_stride], host memory, fp32 (row-major [N][L]: strides (L, 1); the transposed
// copy the Python caller makes: (1, N)).  Rhat is the correctly rounded fp32 mean of the column (float64 accumulation, one
// rounding); torch's own fp32 summation order depends on the host's vector width and may differ from it in the last bits.
// rhat_in (optional, [L], NaN = not given): a column mean supplied by the caller is used INSTEAD of the computed one --
// the Python caller passes torch's own `losses.mean()` for every column whose decision a last-bit change could flip, so that
// lambda-hat is the reference's on the same host (core/calibration/calibrate_model.py scan_loss_table).
// Outputs: *stop_index = the column the scan stopped at (L when it never stopped, i.e. visited all and fell through -> 0
// is reported with *stopped = 0), *lhat, *visited = number of lambdas evaluated; rhat[L] / rhat_plus[L] (optional) receive
// the values of the visited columns (others untouched).
extern "C" int im2im_rcps_scan(const float* table, int64_t N, int32_t L, int64_t row_stride, int64_t col_stride,
                               const float* lambdas, double alpha, double delta, int32_t maxiters, int32_t* stop_index,
                               int32_t* stopped, float* lhat, int32_t* visited, float* rhat, double* rhat_plus,
                               const float* rhat_in) {
  if (!table || !lambdas || !stop_index || !lhat || N <= 0 || L < 2) return -1;
  const float dlambda = lambdas[1] - lambdas[0];
  float lh = lambdas[L - 1] + dlambda;
  lh = lh - 1e-9f;
  const float alpha32 = (float)alpha;
  int32_t stop = 0, did_stop = 0, count = 0;
  for (int32_t j = L - 1; j >= 0; --j) {
    const float* col = table + (int64_t)j * col_stride;
    float r;
    if (rhat_in && rhat_in[j] == rhat_in[j]) {
      r = rhat_in[j];
    } else {
      double s = 0.0;
      for (int64_t i = 0; i < N; ++i) s += (double)col[i * row_stride];
      r = (float)(s / (double)N);
    }
    const double rp = im2im_hb_mu_plus((double)r, N, delta, maxiters);
    if (rhat) rhat[j] = r;
    if (rhat_plus) rhat_plus[j] = rp;
    ++count;
    stop = j;
    if (r >= alpha32 || rp > alpha) { lh = lambdas[j]; did_stop = 1; break; }
  }
  *stop_index = stop;
  if (stopped) *stopped = did_stop;
  *lhat = lh;
  if (visited) *visited = count;
  return 0;
}
