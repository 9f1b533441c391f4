// Host-side numerics of the product library (see host_tables.h).
// Reference semantics restated: include/fe_time.h:157-305, 351-409, 485-514, 643-744;
// fe_time.cc:152-169; include/operators.h:870-965.  deal.II rules (QGauss, QGaussLobatto,
// QGaussRadau) are rebuilt from their mathematical definition.
#include "host_tables.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <random>
#include <stdexcept>

namespace stfem {

namespace {

// Legendre P_n and P_n' on [-1,1] by the three-term recurrence
struct LegVal { double p, dp, pm1; };
LegVal legendre_eval(int n, double x)
{
  double a = 1.0, b = x;
  if (n == 0) return {1.0, 0.0, 0.0};
  for (int k = 1; k < n; ++k) {
    const double c = ((2 * k + 1) * x * b - k * a) / (k + 1);
    a = b;
    b = c;
  }
  const double dp = (std::abs(std::abs(x) - 1.0) < 1e-300) ? 0.0 : n * (x * b - a) / (x * x - 1.0);
  return {b, dp, a};
}

// monomial coefficients (ascending) of P_n
std::vector<double> legendre_monomials(int n)
{
  std::vector<double> a{1.0}, b{0.0, 1.0};
  if (n == 0) return a;
  for (int k = 1; k < n; ++k) {
    std::vector<double> c(k + 2, 0.0);
    for (int i = 0; i <= k; ++i) c[i + 1] += (2.0 * k + 1) / (k + 1) * b[i];
    for (int i = 0; i < k; ++i) c[i] -= double(k) / (k + 1) * a[i];
    a = b;
    b = c;
  }
  return b;
}

// all (real, simple) roots of a polynomial with ascending coefficients: Aberth-Ehrlich
std::vector<double> real_roots(std::vector<double> c)
{
  while (!c.empty() && c.back() == 0.0) c.pop_back();
  const int n = int(c.size()) - 1;
  std::vector<std::complex<double>> z(n);
  for (int i = 0; i < n; ++i) z[i] = std::polar(0.9, 2.0 * M_PI * i / n + 0.4);
  auto eval = [&](std::complex<double> x, std::complex<double> &d) {
    std::complex<double> p = c[n];
    d = 0.0;
    for (int i = n - 1; i >= 0; --i) {
      d = d * x + p;
      p = p * x + c[i];
    }
    return p;
  };
  for (int it = 0; it < 500; ++it) {
    double change = 0.0;
    for (int i = 0; i < n; ++i) {
      std::complex<double> d, p = eval(z[i], d);
      std::complex<double> s = 0.0;
      for (int j = 0; j < n; ++j)
        if (j != i) s += 1.0 / (z[i] - z[j]);
      const std::complex<double> w = p / d;
      const std::complex<double> dz = w / (1.0 - w * s);
      z[i] -= dz;
      change = std::max(change, std::abs(dz));
    }
    if (change < 1e-15) break;
  }
  std::vector<double> r(n);
  for (int i = 0; i < n; ++i) r[i] = z[i].real();
  std::sort(r.begin(), r.end());
  return r;
}

std::vector<double> poly_deriv(const std::vector<double> &c)
{
  std::vector<double> d(c.size() > 1 ? c.size() - 1 : 1, 0.0);
  for (size_t i = 1; i < c.size(); ++i) d[i - 1] = i * c[i];
  return d;
}

template <class F, class DF> double newton_polish(double x, F f, DF df)
{
  for (int it = 0; it < 50; ++it) {
    const double dx = f(x) / df(x);
    x -= dx;
    if (std::abs(dx) < 1e-17) break;
  }
  return x;
}

void symmetrise(std::vector<double> &r)
{
  const int n = int(r.size());
  for (int i = 0; i < n / 2; ++i) {
    const double v = 0.5 * (r[n - 1 - i] - r[i]);
    r[i] = -v;
    r[n - 1 - i] = v;
  }
  if (n & 1) r[n / 2] = 0.0;
}

} // namespace

void gauss_rule(int n, std::vector<double> &x, std::vector<double> &w)
{
  std::vector<double> r = real_roots(legendre_monomials(n));
  for (double &v : r)
    v = newton_polish(
      v, [&](double t) { return legendre_eval(n, t).p; }, [&](double t) { return legendre_eval(n, t).dp; });
  symmetrise(r);
  x.resize(n);
  w.resize(n);
  for (int i = 0; i < n; ++i) {
    const double d = legendre_eval(n, r[i]).dp;
    x[i] = 0.5 * (r[i] + 1.0);
    w[i] = 1.0 / ((1.0 - r[i] * r[i]) * d * d);
  }
}

std::vector<double> lobatto_points(int n)
{
  std::vector<double> x(n);
  x[0] = 0.0;
  x[n - 1] = 1.0;
  if (n > 2) {
    const int N = n - 1;
    // interior nodes: roots of P_N'
    std::vector<double> r = real_roots(poly_deriv(legendre_monomials(N)));
    for (double &v : r)
      v = newton_polish(
        v, [&](double t) { return legendre_eval(N, t).dp; },
        [&](double t) { // P_N'' from the Legendre ODE
          const LegVal l = legendre_eval(N, t);
          return (2.0 * t * l.dp - N * (N + 1.0) * l.p) / (1.0 - t * t);
        });
    symmetrise(r);
    for (int i = 0; i < n - 2; ++i) x[i + 1] = 0.5 * (r[i] + 1.0);
  }
  return x;
}

std::vector<double> radau_right_points(int n)
{
  std::vector<double> x(n);
  x[n - 1] = 1.0;
  if (n > 1) {
    // P_n - P_{n-1} has the root +1; deflate it synthetically, then polish on the full form
    std::vector<double> a = legendre_monomials(n), b = legendre_monomials(n - 1);
    for (size_t i = 0; i < b.size(); ++i) a[i] -= b[i];
    std::vector<double> q(n, 0.0); // a(x) = (x-1) q(x)
    double carry = 0.0;
    for (int i = n; i >= 1; --i) {
      q[i - 1] = a[i] + carry;
      carry = q[i - 1];
    }
    std::vector<double> r = real_roots(q);
    for (double &v : r)
      v = newton_polish(
        v, [&](double t) { return legendre_eval(n, t).p - legendre_eval(n - 1, t).p; },
        [&](double t) { return legendre_eval(n, t).dp - legendre_eval(n - 1, t).dp; });
    for (int i = 0; i < n - 1; ++i) x[i] = 0.5 * (r[i] + 1.0);
  }
  return x;
}

void lagrange_tables(const std::vector<double> &nodes, const std::vector<double> &x, Mat &V, Mat &G)
{
  const int n = int(nodes.size()), m = int(x.size());
  V.assign(size_t(m) * n, 0.0);
  G.assign(size_t(m) * n, 0.0);
  std::vector<double> bw(n, 1.0); // barycentric weights
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b)
      if (a != b) bw[a] /= (nodes[a] - nodes[b]);
  for (int q = 0; q < m; ++q)
    for (int a = 0; a < n; ++a) {
      double val = bw[a], der = 0.0;
      for (int b = 0; b < n; ++b)
        if (b != a) val *= (x[q] - nodes[b]);
      for (int c = 0; c < n; ++c) {
        if (c == a) continue;
        double t = bw[a];
        for (int b = 0; b < n; ++b)
          if (b != a && b != c) t *= (x[q] - nodes[b]);
        der += t;
      }
      V[size_t(q) * n + a] = val;
      G[size_t(q) * n + a] = der;
    }
}

void eo_pack(int n, const Mat &X, double *out)
{
  const int h = n / 2, m = n / 2;
  for (int q = 0; q < h; ++q)
    for (int i = 0; i < h; ++i) {
      out[q * h + i] = 0.5 * (X[q * n + i] + X[q * n + n - 1 - i]);
      out[h * h + q * h + i] = 0.5 * (X[q * n + i] - X[q * n + n - 1 - i]);
    }
  if (n & 1) {
    for (int q = 0; q < h; ++q) out[2 * h * h + q] = X[q * n + m];
    for (int i = 0; i < h; ++i) out[2 * h * h + h + i] = X[m * n + i];
    out[2 * h * h + 2 * h] = X[m * n + m];
  }
}

static Mat transpose(int n, const Mat &A)
{
  Mat T(A.size());
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) T[j * n + i] = A[i * n + j];
  return T;
}

namespace {

// Generalised symmetric eigenproblem K v = lam M v of a small dense pair (m <= 4), M SPD:
// Cholesky M = L L^T, cyclic Jacobi on L^-1 K L^-T.  Returns eigenvalues (ascending) and the
// M-orthonormal eigenvectors as columns of V (row-major m x m).
void small_gen_eig(int m, Mat K, Mat M, std::vector<double> &lam, Mat &V)
{
  Mat L(size_t(m) * m, 0.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = M[i * m + j];
      for (int k = 0; k < j; ++k) s -= L[i * m + k] * L[j * m + k];
      L[i * m + j] = i == j ? std::sqrt(s) : s / L[j * m + j];
    }
  // Li = L^-1 (lower triangular)
  Mat Li(size_t(m) * m, 0.0);
  for (int c = 0; c < m; ++c)
    for (int i = c; i < m; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= L[i * m + k] * Li[k * m + c];
      Li[i * m + c] = s / L[i * m + i];
    }
  // A = Li K Li^T
  Mat T(size_t(m) * m, 0.0), A(size_t(m) * m, 0.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j)
      for (int k = 0; k < m; ++k) T[i * m + j] += Li[i * m + k] * K[k * m + j];
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j)
      for (int k = 0; k < m; ++k) A[i * m + j] += T[i * m + k] * Li[j * m + k];
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < i; ++j) A[i * m + j] = A[j * m + i] = 0.5 * (A[i * m + j] + A[j * m + i]);
  Mat Q(size_t(m) * m, 0.0);
  for (int i = 0; i < m; ++i) Q[i * m + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) (i == j ? dia : off) += A[i * m + j] * A[i * m + j];
    if (off <= 1e-34 * dia) break;
    for (int pi = 0; pi < m; ++pi)
      for (int qi = pi + 1; qi < m; ++qi) {
        if (A[pi * m + qi] == 0.0) continue;
        const double theta = (A[qi * m + qi] - A[pi * m + pi]) / (2.0 * A[pi * m + qi]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < m; ++k) { // columns p, q
          const double akp = A[k * m + pi], akq = A[k * m + qi];
          A[k * m + pi] = c * akp - s * akq;
          A[k * m + qi] = s * akp + c * akq;
        }
        for (int k = 0; k < m; ++k) { // rows p, q
          const double apk = A[pi * m + k], aqk = A[qi * m + k];
          A[pi * m + k] = c * apk - s * aqk;
          A[qi * m + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < m; ++k) {
          const double qkp = Q[k * m + pi], qkq = Q[k * m + qi];
          Q[k * m + pi] = c * qkp - s * qkq;
          Q[k * m + qi] = s * qkp + c * qkq;
        }
      }
  }
  std::vector<int> order(m);
  for (int i = 0; i < m; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return A[a * m + a] < A[b * m + b]; });
  lam.resize(m);
  V.assign(size_t(m) * m, 0.0);
  for (int e = 0; e < m; ++e) {
    const int o = order[e];
    lam[e] = A[o * m + o];
    for (int i = 0; i < m; ++i) { // v = L^-T q
      double s = 0.0;
      for (int k = 0; k < m; ++k) s += Li[k * m + i] * Q[k * m + o];
      V[i * m + e] = s;
    }
  }
}

// fills t.fd_W / t.fd_lam (see host_tables.h)
void make_fast_diagonalisation(ShapeTables &t)
{
  const int n = t.n, h = n / 2, ne = n - h;
  Mat M1(size_t(n) * n, 0.0), K1(size_t(n) * n, 0.0);
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < n; ++b)
      for (int q = 0; q < n; ++q) {
        M1[a * n + b] += t.wq[q] * t.S[q * n + a] * t.S[q * n + b];
        K1[a * n + b] += t.wq[q] * t.D[q * n + a] * t.D[q * n + b];
      }
  std::fill(t.fd_W, t.fd_W + EO_MAX, 0.0);
  std::fill(t.fd_lam, t.fd_lam + 8, 0.0);
  // parity bases: even e_i = (d_i + d_{n-1-i})/sqrt2 (i < h), middle d_h; odd o_i = (d_i - d_{n-1-i})/sqrt2
  const double r2 = std::sqrt(0.5);
  for (int parity = 0; parity < 2; ++parity) {
    const int m = parity == 0 ? ne : h;
    if (m == 0) continue;
    Mat B(size_t(n) * m, 0.0); // basis vectors as columns
    for (int i = 0; i < m; ++i) {
      if (parity == 0 && i == h) B[h * m + i] = 1.0; // middle node (odd n only)
      else {
        B[i * m + i] = r2;
        B[(n - 1 - i) * m + i] = parity == 0 ? r2 : -r2;
      }
    }
    Mat Mr(size_t(m) * m, 0.0), Kr(size_t(m) * m, 0.0);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j)
        for (int a = 0; a < n; ++a)
          for (int b = 0; b < n; ++b) {
            Mr[i * m + j] += B[a * m + i] * M1[a * n + b] * B[b * m + j];
            Kr[i * m + j] += B[a * m + i] * K1[a * n + b] * B[b * m + j];
          }
    std::vector<double> lam;
    Mat Vr;
    small_gen_eig(m, Kr, Mr, lam, Vr);
    for (int e = 0; e < m; ++e) {
      // full eigenvector v = B vr, row of W = (M1 v)^T
      std::vector<double> v(n, 0.0), w(n, 0.0);
      for (int a = 0; a < n; ++a)
        for (int i = 0; i < m; ++i) v[a] += B[a * m + i] * Vr[i * m + e];
      for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b) w[a] += M1[a * n + b] * v[b];
      // sign convention: first nonzero entry positive (deterministic tables)
      const double sg = w[0] < 0 ? -1.0 : 1.0;
      if (parity == 0) {
        for (int i = 0; i < ne; ++i) t.fd_W[e * ne + i] = sg * w[i];
        t.fd_lam[e] = lam[e];
      } else {
        for (int i = 0; i < h; ++i) t.fd_W[ne * ne + e * h + i] = sg * w[i];
        t.fd_lam[ne + e] = lam[e];
      }
    }
  }
}

} // namespace

ShapeTables make_shape_tables(int p)
{
  ShapeTables t;
  const int n = p + 1;
  t.n = n;
  t.nodes = lobatto_points(n);
  gauss_rule(n, t.xq, t.wq);
  lagrange_tables(t.nodes, t.xq, t.S, t.D);
  Mat dummy;
  lagrange_tables(t.xq, t.xq, dummy, t.Dcol);
  Mat Si(n * n), Dc(n * n);
  for (int q = 0; q < n; ++q)
    for (int a = 0; a < n; ++a) {
      Si[q * n + a] = std::sqrt(t.wq[q]) * t.S[q * n + a];
      Dc[q * n + a] = std::sqrt(t.wq[q]) * t.Dcol[q * n + a] / std::sqrt(t.wq[a]);
    }
  eo_pack(n, Si, t.eo_Si);
  eo_pack(n, transpose(n, Si), t.eo_SiT);
  eo_pack(n, Dc, t.eo_Dc);
  eo_pack(n, transpose(n, Dc), t.eo_DcT);
  {
    Mat Lm(n * n, 0.0);
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b)
        for (int q = 0; q < n; ++q) Lm[a * n + b] += Dc[q * n + a] * Dc[q * n + b];
    eo_pack(n, Lm, t.eo_L);
  }
  eo_pack(n, t.S, t.eo_S);
  eo_pack(n, transpose(n, t.S), t.eo_ST);
  eo_pack(n, t.Dcol, t.eo_Dq);
  eo_pack(n, transpose(n, t.Dcol), t.eo_DqT);
  make_fast_diagonalisation(t);
  return t;
}

// --------------------------------------------------------------------------- temporal matrices

int time_nb(int type, int r, int nsteps) { return (type == 0 ? r : r + 1) * nsteps; }

namespace {

struct Step1 {
  int nt;
  Mat A, B;               // tau*M_t (lhs part), D_t (lhs part)
  std::vector<double> t2, t3; // the reference's tmp[2], tmp[3] after the CGP/DG branch
  std::vector<double> G, Z;   // returned Gamma, Zeta of one step
};

// get_cg_weights / get_dg_weights + split_lhs_rhs + the branch at fe_time.h:359-371
Step1 single_step(int type, int r, double tau)
{
  Step1 s;
  std::vector<double> xq, wq;
  gauss_rule(r + 2, xq, wq);
  if (type == 0) {
    if (r < 1) throw std::invalid_argument("cG needs r >= 1");
    const std::vector<double> trial = lobatto_points(r + 1);
    const std::vector<double> test(trial.begin() + 1, trial.end());
    Mat Vt, Gt, Vs, Gs;
    lagrange_tables(trial, xq, Vt, Gt);
    lagrange_tables(test, xq, Vs, Gs);
    s.nt = r;
    s.A.assign(r * r, 0.0);
    s.B.assign(r * r, 0.0);
    s.t2.assign(r, 0.0);
    s.t3.assign(r, 0.0);
    for (int i = 0; i < r; ++i)
      for (int j = 0; j <= r; ++j) {
        double m = 0, d = 0;
        for (size_t q = 0; q < xq.size(); ++q) {
          m += wq[q] * Vs[q * r + i] * Vt[q * (r + 1) + j];
          d += wq[q] * Vs[q * r + i] * Gt[q * (r + 1) + j];
        }
        if (j == 0) {
          s.t2[i] = -tau * m;
          s.t3[i] = -d;
        } else {
          s.A[i * r + j - 1] = tau * m;
          s.B[i * r + j - 1] = d;
        }
      }
    s.G = s.t2;
    s.Z = s.t3;
  } else {
    if (r < 0) throw std::invalid_argument("dG needs r >= 0");
    const int nt = r + 1;
    const std::vector<double> pts = radau_right_points(nt);
    Mat V, G, V0, G0;
    lagrange_tables(pts, xq, V, G);
    lagrange_tables(pts, std::vector<double>{0.0}, V0, G0);
    s.nt = nt;
    s.A.assign(nt * nt, 0.0);
    s.B.assign(nt * nt, 0.0);
    for (int i = 0; i < nt; ++i)
      for (int j = 0; j < nt; ++j) {
        double m = 0, d = V0[i] * V0[j];
        for (size_t q = 0; q < xq.size(); ++q) {
          m += wq[q] * V[q * nt + i] * V[q * nt + j];
          d += wq[q] * V[q * nt + i] * G[q * nt + j];
        }
        s.A[i * nt + j] = tau * m;
        s.B[i * nt + j] = d;
      }
    s.t2.assign(nt, 0.0);
    s.t3.assign(V0.begin(), V0.begin() + nt);
    s.G = s.t3; // DG: returned Gamma = jump, Zeta = 0
    s.Z = s.t2;
  }
  return s;
}

Mat matmul(int m, int k, int n, const Mat &A, const Mat &B)
{
  Mat C(size_t(m) * n, 0.0);
  for (int i = 0; i < m; ++i)
    for (int l = 0; l < k; ++l)
      for (int j = 0; j < n; ++j) C[i * n + j] += A[i * k + l] * B[l * n + j];
  return C;
}

Mat inverse(int n, Mat A)
{
  Mat I(size_t(n) * n, 0.0);
  for (int i = 0; i < n; ++i) I[i * n + i] = 1.0;
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::abs(A[r * n + c]) > std::abs(A[piv * n + c])) piv = r;
    for (int j = 0; j < n; ++j) {
      std::swap(A[c * n + j], A[piv * n + j]);
      std::swap(I[c * n + j], I[piv * n + j]);
    }
    const double d = A[c * n + c];
    for (int j = 0; j < n; ++j) {
      A[c * n + j] /= d;
      I[c * n + j] /= d;
    }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = A[r * n + c];
      for (int j = 0; j < n; ++j) {
        A[r * n + j] -= f * A[c * n + j];
        I[r * n + j] -= f * I[c * n + j];
      }
    }
  }
  return I;
}

} // namespace

// ---- time-multigrid transfer matrices (reference include/fe_time.h:749-898) ----
// The deal.II pieces the reference calls, spelled out for the 1D Lagrange elements on the Gauss-Lobatto (cG) /
// right Gauss-Radau (dG) points: get_prolongation_matrix(child) = the parent basis at the child's support points;
// FE_Q::get_restriction_matrix(child) = the child basis at the parent's support points lying in that child;
// FE_DGQArbitraryNodes::get_restriction_matrix and FETools::get_projection_matrix = L2 projections.
namespace {
std::vector<double> time_nodes(int type, int r) { return type == 0 ? lobatto_points(r + 1) : radau_right_points(r + 1); }
Mat lagrange_at(const std::vector<double> &nodes, const std::vector<double> &x)
{
  Mat V, G;
  lagrange_tables(nodes, x, V, G);
  return V; // [x][node]
}
Mat mass_like(const std::vector<double> &rows, const std::vector<double> &cols, int nq)
{
  std::vector<double> xq, wq;
  gauss_rule(nq, xq, wq);
  const Mat Vr = lagrange_at(rows, xq), Vc = lagrange_at(cols, xq);
  const int nr = int(rows.size()), nc = int(cols.size());
  Mat M(size_t(nr) * nc, 0.0);
  for (int q = 0; q < nq; ++q)
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) M[i * nc + j] += wq[q] * Vr[q * nr + i] * Vc[q * nc + j];
  return M;
}
} // namespace

int time_prolongation(int type, int r, int nsteps, Mat &out, int &m, int &n)
{
  if ((type != 0 && type != 1) || r < (type == 0 ? 1 : 0) || r > 8 || nsteps < 2 || (nsteps & (nsteps - 1))) return -1;
  const std::vector<double> x = time_nodes(type, r);
  const int np = r + 1, skip = type == 0 ? 1 : 0, nd = np - skip; // cG: the left end belongs to the previous step
  std::vector<double> xl(np), xr(np);
  for (int i = 0; i < np; ++i) { xl[i] = x[i] / 2; xr[i] = (x[i] + 1) / 2; }
  const Mat L = lagrange_at(x, xl), R = lagrange_at(x, xr);
  m = nd * nsteps; n = nd * nsteps / 2;
  out.assign(size_t(m) * n, 0.0);
  for (int it = 0; it < nsteps / 2; ++it)
    for (int i = 0; i < nd; ++i)
      for (int j = 0; j < nd; ++j) {
        out[size_t(2 * nd * it + i) * n + nd * it + j] = L[(i + skip) * np + j + skip];
        out[size_t(2 * nd * it + nd + i) * n + nd * it + j] = R[(i + skip) * np + j + skip];
      }
  return 0;
}

int time_restriction(int type, int r, int nsteps, Mat &out, int &m, int &n)
{
  if ((type != 0 && type != 1) || r < (type == 0 ? 1 : 0) || r > 8 || nsteps < 2 || (nsteps & (nsteps - 1))) return -1;
  const std::vector<double> x = time_nodes(type, r);
  const int np = r + 1, skip = type == 0 ? 1 : 0, nd = np - skip;
  Mat Rl(size_t(np) * np, 0.0), Rr(size_t(np) * np, 0.0);
  if (type == 0) {
    const double eps = 1e-12;
    for (int i = 0; i < np; ++i) {
      if (x[i] <= 0.5 + eps) {
        const Mat v = lagrange_at(x, {std::min(1.0, 2 * x[i])});
        for (int j = 0; j < np; ++j) Rl[i * np + j] = v[j];
      }
      if (x[i] >= 0.5 - eps) {
        const Mat v = lagrange_at(x, {std::max(0.0, 2 * x[i] - 1)});
        for (int j = 0; j < np; ++j) Rr[i * np + j] = v[j];
      }
    }
  } else {
    std::vector<double> xl(np), xr(np);
    for (int i = 0; i < np; ++i) { xl[i] = x[i] / 2; xr[i] = (x[i] + 1) / 2; }
    const Mat Pl = lagrange_at(x, xl), Pr = lagrange_at(x, xr), M = mass_like(x, x, r + 2), Mi = inverse(np, M);
    auto half_projection = [&](const Mat &P) { // M^-1 P^T M / 2
      Mat PT(size_t(np) * np);
      for (int i = 0; i < np; ++i)
        for (int j = 0; j < np; ++j) PT[i * np + j] = P[j * np + i];
      Mat A = matmul(np, np, np, Mi, matmul(np, np, np, PT, M));
      for (double &v : A) v *= 0.5;
      return A;
    };
    Rl = half_projection(Pl);
    Rr = half_projection(Pr);
  }
  m = nd * nsteps / 2; n = nd * nsteps;
  out.assign(size_t(m) * n, 0.0);
  for (int it = 0; it < nsteps / 2; ++it)
    for (int i = 0; i < nd; ++i)
      for (int j = 0; j < nd; ++j) {
        out[size_t(nd * it + i) * n + 2 * nd * it + j] = Rl[(i + skip) * np + j + skip];
        out[size_t(nd * it + i) * n + 2 * nd * it + nd + j] = Rr[(i + skip) * np + j + skip];
      }
  return 0;
}

int time_projection(int type, int r_src, int r_dst, int nsteps, Mat &out, int &m, int &n)
{
  const int rmin = type == 0 ? 1 : 0;
  if ((type != 0 && type != 1) || r_src < rmin || r_dst < rmin || r_src > 8 || r_dst > 8 || nsteps < 1) return -1;
  const std::vector<double> xs = time_nodes(type, r_src), xd = time_nodes(type, r_dst);
  const int ns = r_src + 1, nd = r_dst + 1, nq = std::max(r_src, r_dst) + 2;
  const Mat P = matmul(nd, nd, ns, inverse(nd, mass_like(xd, xd, nq)), mass_like(xd, xs, nq)); // [dst][src]
  if (type == 1) {
    m = nsteps * nd; n = nsteps * ns;
    out.assign(size_t(m) * n, 0.0);
    for (int it = 0; it < nsteps; ++it)
      for (int i = 0; i < nd; ++i)
        for (int j = 0; j < ns; ++j) out[size_t(it * nd + i) * n + it * ns + j] = P[i * ns + j];
    return 0;
  }
  // cG: consecutive steps share their end point (a later step's block overwrites the shared entry), then the
  // first row and column - the dof at the start of the slab - are dropped
  const int fm = nsteps * r_dst + 1, fn = nsteps * r_src + 1;
  Mat full(size_t(fm) * fn, 0.0);
  for (int it = 0; it < nsteps; ++it)
    for (int i = 0; i < nd; ++i)
      for (int j = 0; j < ns; ++j) full[size_t(it * r_dst + i) * fn + it * r_src + j] = P[i * ns + j];
  m = fm - 1; n = fn - 1;
  out.assign(size_t(m) * n, 0.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) out[size_t(i) * n + j] = full[size_t(i + 1) * fn + j + 1];
  return 0;
}

int fe_time_weights(int type, int r, double tau, int nsteps, Mat &Alpha, Mat &Beta, Mat &Gamma,
                    Mat &Zeta)
{
  const Step1 s = single_step(type, r, tau);
  const int nt = s.nt, nb = nt * nsteps;
  Alpha.assign(size_t(nb) * nb, 0.0);
  Beta.assign(size_t(nb) * nb, 0.0);
  Gamma.assign(nb, 0.0);
  Zeta.assign(nb, 0.0);
  for (int it = 0; it < nsteps; ++it) {
    const int o = it * nt;
    for (int i = 0; i < nt; ++i)
      for (int j = 0; j < nt; ++j) {
        Alpha[(o + i) * nb + o + j] = s.A[i * nt + j];
        Beta[(o + i) * nb + o + j] = s.B[i * nt + j];
      }
    if (it + 1 < nsteps) // last DoF of step `it` feeds step it+1 (fe_time.h:383-391)
      for (int j = 0; j < nt; ++j) {
        Alpha[(o + nt + j) * nb + o + nt - 1] = -s.t2[j];
        Beta[(o + nt + j) * nb + o + nt - 1] = -s.t3[j];
      }
  }
  for (int i = 0; i < nt; ++i) {
    Gamma[i] = s.G[i];
    Zeta[i] = s.Z[i];
  }
  return nb;
}

int fe_time_weights_wave(int type, int r, double tau, int nsteps, Mat &AL, Mat &BL, Mat &uK,
                         Mat &uM, Mat &vM)
{
  const Step1 s = single_step(type, r, tau);
  const int nt = s.nt, nb = nt * nsteps;
  const Mat &A = s.A, &B = s.B;
  const std::vector<double> &G = s.G, &Z = s.Z;
  const Mat BAi = matmul(nt, nt, nt, B, inverse(nt, A));
  const Mat BAB = matmul(nt, nt, nt, BAi, B);
  const Mat BAG = matmul(nt, nt, 1, BAi, G);
  const double all = A[(nt - 1) * nt + nt - 1];
  const double gxai = G[nt - 1] / all;
  AL.assign(size_t(nb) * nb, 0.0);
  BL.assign(size_t(nb) * nb, 0.0);
  uK.assign(nb, 0.0);
  uM.assign(nb, 0.0);
  vM.assign(nb, 0.0);
  auto al = [&](int i, int j) -> double & { return AL[size_t(i) * nb + j]; };
  auto bl = [&](int i, int j) -> double & { return BL[size_t(i) * nb + j]; };
  const double *Blast = &B[(nt - 1) * nt];
  if (type == 0) {
    const Mat BAZ = matmul(nt, nt, 1, BAi, Z);
    std::vector<double> ZmBAG(nt);
    for (int i = 0; i < nt; ++i) ZmBAG[i] = Z[i] - BAG[i];
    const double zxai = Z[nt - 1] / all;
    for (int it = 0; it < nsteps; ++it)
      for (int jt = 0; jt <= it; ++jt)
        for (int i = 0; i < nt; ++i) {
          const int row = i + it * nt;
          if (it == 0 && jt == 0) {
            uK[i] = G[i];
            uM[i] = BAZ[i];
            vM[i] = ZmBAG[i];
          } else if (jt == 0) {
            uM[row] = -zxai * std::pow(gxai, it - 1) * ZmBAG[i];
            vM[row] = std::pow(gxai, it) * ZmBAG[i];
          }
          if (it == jt + 1) {
            al(row, nt - 1 + jt * nt) = -G[i];
            bl(row, nt - 1 + jt * nt) = -BAZ[i];
          }
          if (it == jt) {
            for (int j = 0; j < nt; ++j) {
              al(row, j + it * nt) = A[i * nt + j];
              bl(row, j + it * nt) = BAB[i * nt + j];
            }
          } else {
            for (int j = 0; j < nt; ++j) {
              const double zmbab = ZmBAG[i] * Blast[j] / all;
              double v = -std::pow(gxai, it - jt - 1) * zmbab;
              if (it > 1 && it - 1 > jt && j == nt - 1)
                v += std::pow(gxai, it - jt - 2) * zxai * ZmBAG[i];
              bl(row, j + jt * nt) += v;
            }
          }
        }
  } else {
    for (int it = 0; it < nsteps; ++it)
      for (int i = 0; i < nt; ++i) {
        if (it == 0) {
          uM[i] = BAG[i];
          vM[i] = G[i];
        }
        if (it == 1) uM[nt + i] = -G[i] * gxai;
        if (it + 1 < nsteps)
          for (int j = 0; j < nt; ++j)
            bl(j + (it + 1) * nt, i + it * nt) =
              -G[j] * Blast[i] / all - (i == nt - 1 ? BAG[j] : 0.0);
        if (it + 2 < nsteps && i == nt - 1)
          for (int j = 0; j < nt; ++j) bl(j + (it + 2) * nt, i + it * nt) = G[j] * gxai;
        for (int j = 0; j < nt; ++j) {
          al(i + it * nt, j + it * nt) = A[i * nt + j];
          bl(i + it * nt, j + it * nt) = BAB[i * nt + j];
        }
      }
  }
  return nb;
}

// --------------------------------------------------------------------------- mesh / coefficient

void mesh_vertices(const int32_t gn[3], const double lo[3], const double up[3], double distort,
                   uint64_t seed, int32_t z0, int32_t z1, double *out)
{
  const double h[3] = {(up[0] - lo[0]) / gn[0], (up[1] - lo[1]) / gn[1], (up[2] - lo[2]) / gn[2]};
  std::mt19937_64 rng(seed);
  const int64_t per_plane = int64_t(gn[0] + 1) * (gn[1] + 1);
  if (distort != 0.0) rng.discard(uint64_t(3) * per_plane * z0);
  auto u11 = [&]() { return double(rng() >> 11) * (2.0 / 9007199254740992.0) - 1.0; };
  int64_t o = 0;
  for (int k = z0; k <= z1; ++k)
    for (int j = 0; j <= gn[1]; ++j)
      for (int i = 0; i <= gn[0]; ++i, ++o) {
        double d[3] = {0, 0, 0};
        if (distort != 0.0) {
          d[0] = u11();
          d[1] = u11();
          d[2] = u11();
          const bool interior = i > 0 && i < gn[0] && j > 0 && j < gn[1] && k > 0 && k < gn[2];
          if (!interior) d[0] = d[1] = d[2] = 0.0;
        }
        out[3 * o + 0] = lo[0] + h[0] * i + distort * h[0] * d[0];
        out[3 * o + 1] = lo[1] + h[1] * j + distort * h[1] * d[1];
        out[3 * o + 2] = lo[2] + h[2] * k + distort * h[2] * d[2];
      }
}

void coefficient_per_cell(const int32_t nc[3], const double *v, double c1, double c2, double c3,
                          double distort, const int32_t sub[3], const double lo[3],
                          const double up[3], double *out)
{
  std::vector<double> table;
  double step[3] = {1, 1, 1};
  if (distort != 0.0) {
    // boost::random::mt19937(default_seed) + uniform_real_distribution(1-d, 1+d):
    // one 32-bit draw per value, value = draw / 2^32 * (max-min) + min  (operators.h:905-921)
    std::mt19937 rng(5489u);
    table.resize(size_t(sub[0]) * sub[1] * sub[2]);
    for (double &t : table) {
      double res;
      do {
        res = double(rng()) / 4294967296.0 * (2.0 * distort) + (1.0 - distort);
      } while (!(res < 1.0 + distort));
      t = res;
    }
    for (int d = 0; d < 3; ++d) step[d] = (up[d] - lo[d]) / sub[d];
  }
  const int64_t nvx = nc[0] + 1, nvy = nc[1] + 1;
  int64_t cell = 0;
  for (int cz = 0; cz < nc[2]; ++cz)
    for (int cy = 0; cy < nc[1]; ++cy)
      for (int cx = 0; cx < nc[0]; ++cx, ++cell) {
        double c[3] = {0, 0, 0};
        for (int k = 0; k < 2; ++k)
          for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 2; ++i) {
              const int64_t idx = (cx + i) + nvx * ((cy + j) + nvy * int64_t(cz + k));
              for (int d = 0; d < 3; ++d) c[d] += 0.125 * v[3 * idx + d];
            }
        double val = c1;
        if (c[1] >= 0.2) val = (c[0] < 0.2) ? c2 : c3;
        if (!table.empty()) {
          const unsigned ix = unsigned((c[0] - lo[0]) / step[0]);
          const unsigned iy = unsigned((c[1] - lo[1]) / step[1]);
          const unsigned iz = unsigned((c[2] - lo[2]) / step[2]);
          val *= table[(size_t(ix) * sub[1] + iy) * sub[2] + iz]; // Table<3>::fill, last index fastest
        }
        out[cell] = val;
      }
}

// ---- level schedule of the space-time multigrid ----

std::vector<int> poly_mg_sequence(int k_max, int k_min, int sequence_type)
{
  std::vector<int> degrees{k_max};
  while (degrees.back() > k_min) {
    const int prev = degrees.back();
    if (sequence_type == 0) degrees.push_back(prev / 2);
    else if (sequence_type == 1) degrees.push_back(prev - 1);
    else if (sequence_type == 2) degrees.push_back(k_min);
    else return {};
  }
  std::reverse(degrees.begin(), degrees.end());
  return degrees;
}

std::string mg_sequence(int n_sp_lvl, int n_k, int n_p, int n_timesteps_at_once, int n_timesteps_at_once_min, char lower_lvl,
                        int coarsening_type, bool time_before_space, bool use_p_multigrid_space, bool zip_from_back)
{
  int n_tau = 0;
  for (int q = n_timesteps_at_once / n_timesteps_at_once_min; q > 1; q /= 2) ++n_tau;
  const int n_kl = n_k - 1, n_pl = use_p_multigrid_space ? n_p - 1 : 0, n_hl = n_sp_lvl - 1;
  const bool k_low = lower_lvl == 'k';
  // the lower kind of each family comes first (= nearer the coarse end)
  const std::string time_levels = k_low ? std::string(n_kl, 'k') + std::string(n_tau, 't') : std::string(n_tau, 't') + std::string(n_kl, 'k');
  const std::string space_levels = k_low ? std::string(n_pl, 'p') + std::string(n_hl, 'h') : std::string(n_hl, 'h') + std::string(n_pl, 'p');
  const std::string &a = time_before_space ? time_levels : space_levels, &b = time_before_space ? space_levels : time_levels;
  std::string seq;
  if (coarsening_type == 0) { // space_or_time: one family after the other
    if (zip_from_back) seq = std::string(a.rbegin(), a.rend()) + std::string(b.rbegin(), b.rend());
    else seq = a + b;
    return seq;
  }
  // space_and_time: interleaved, zipped from the fine end if zip_from_back
  const size_t n = std::max(a.size(), b.size());
  for (size_t i = 0; i < n; ++i) {
    if (i < a.size()) seq.push_back(zip_from_back ? a[a.size() - 1 - i] : a[i]);
    if (i < b.size()) seq.push_back(zip_from_back ? b[b.size() - 1 - i] : b[i]);
  }
  if (zip_from_back) std::reverse(seq.begin(), seq.end());
  return seq;
}

std::vector<int> precondition_stmg_types(const std::string &seq, int coarsening_type, bool time_before_space, int smoother)
{
  std::vector<int> ret(seq.size() + 1, smoother);
  if (coarsening_type == 0 || seq.empty()) return ret;
  auto space = [](char c) { return c == 'h' || c == 'p'; };
  // of a (space, time) pair of transfers that together form one space-time coarsening only the first level smooths
  for (size_t i = 0; i + 1 < seq.size(); ++i)
    if (time_before_space ? (space(seq[i]) && !space(seq[i + 1])) : (!space(seq[i]) && space(seq[i + 1]))) {
      ret[i] = smoother;
      ret[i + 1] = 0;
      ++i;
    }
  return ret;
}

} // namespace stfem
