// CPU restatement of Zeth's Groth16 prover hot path -- TEST INFRASTRUCTURE ONLY (the checker and the
// bench's cpu_baseline; never linked into or called by the product library).
//
// The arithmetic of this path lives in clearmatics/libsnark (+ nested libff, libfqfft), an un-vendored
// submodule that is an EMPTY directory in /root/reference (pinned commit unrecoverable), so
// this file restates the published algorithms those libraries implement, anchored on Zeth's call sites:
//   * Fp_model Montgomery arithmetic on 64-bit limbs          (usage: libzeth/mpc/groth16/powersoftau_utils.tcc:21-34)
//   * Jacobian G1/G2 add / mixed_add / dbl                     (libff alt_bn128_g1.cpp etc. [UPSTREAM-MEMORY])
//   * multi_exp<BDLO12>, multi_exp_with_mixed_addition, chunks (libzeth/core/multi_exp.tcc:13-35)
//   * basic_radix2_domain FFT / iFFT / cosetFFT / icosetFFT    (libzeth/core/evaluator_from_lagrange.tcc:43,
//                                                               powersoftau_utils.tcc:125-134)
//   * r1cs_to_qap_witness_map + r1cs_gg_ppzksnark_prover       (libzeth/snarks/groth16/groth16_snark.tcc:36-46)
// Parity status: pinned against oracle/zeth_model.py (itself pinned by the reference's known-answer vectors,
// tests/test_oracle_kats.py) in tests/test_oracle_cpu.py; MSM/NTT/proof outputs have no reference-authored
// vectors ("parity unpinned" at that level).
//
// Differences from the product code are deliberate so that the two are independent: 64-bit limbs with
// unsigned __int128 (the product uses 32-bit limbs), Jacobian coordinates (product: XYZZ), unsigned c-bit
// windows exactly as BDLO12 (product: signed digits + counting sort), in-place bit-reversal DIT FFT.
//
// Build: g++ -O2 -fopenmp -shared -fPIC (oracle/Makefile).  -march=native is NOT used because the library is
// built in one container and executed on another host.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------------------
// prime fields
// ------------------------------------------------------------------------------------------------
template <int N>
struct FieldCtx {
  u64 p[N];
  u64 inv;     // -p^-1 mod 2^64
  u64 one[N];  // R mod p
  u64 r2[N];   // R^2 mod p
};

template <int N>
static bool ge(const u64* a, const u64* b) {
  for (int i = N - 1; i >= 0; i--) {
    if (a[i] != b[i]) return a[i] > b[i];
  }
  return true;
}
template <int N>
static u64 sub_n(u64* r, const u64* a, const u64* b) {
  u64 br = 0;
  for (int i = 0; i < N; i++) {
    u128 t = (u128)a[i] - b[i] - br;
    r[i] = (u64)t;
    br = (u64)(t >> 64) & 1;
  }
  return br;
}
template <int N>
static u64 add_n(u64* r, const u64* a, const u64* b) {
  u64 c = 0;
  for (int i = 0; i < N; i++) {
    u128 t = (u128)a[i] + b[i] + c;
    r[i] = (u64)t;
    c = (u64)(t >> 64);
  }
  return c;
}

template <int N>
static void field_init(FieldCtx<N>& C, const u64* p) {
  memcpy(C.p, p, sizeof(C.p));
  // inv by Newton iteration on 64 bits
  u64 x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
  C.inv = (u64)0 - x;
  // R mod p and R^2 mod p by repeated doubling of 1
  u64 t[N];
  memset(t, 0, sizeof(t));
  t[0] = 1;
  for (int i = 0; i < 2 * 64 * N; i++) {
    u64 d[N];
    u64 c = add_n<N>(d, t, t);
    if (c || ge<N>(d, C.p)) sub_n<N>(d, d, C.p);
    memcpy(t, d, sizeof(t));
    if (i == 64 * N - 1) memcpy(C.one, t, sizeof(t));
  }
  memcpy(C.r2, t, sizeof(t));
}

template <int N, const FieldCtx<N>& C>
struct Fe {
  static constexpr int LIMBS = N;
  u64 v[N];
  static Fe zero() {
    Fe r;
    memset(r.v, 0, sizeof(r.v));
    return r;
  }
  static Fe one() {
    Fe r;
    memcpy(r.v, C.one, sizeof(r.v));
    return r;
  }
  bool is_zero() const {
    u64 o = 0;
    for (int i = 0; i < N; i++) o |= v[i];
    return o == 0;
  }
  bool operator==(const Fe& b) const { return memcmp(v, b.v, sizeof(v)) == 0; }
  bool operator!=(const Fe& b) const { return !(*this == b); }
  Fe operator+(const Fe& b) const {
    Fe r;
    u64 c = add_n<N>(r.v, v, b.v);
    if (c || ge<N>(r.v, C.p)) sub_n<N>(r.v, r.v, C.p);
    return r;
  }
  Fe operator-(const Fe& b) const {
    Fe r;
    if (sub_n<N>(r.v, v, b.v)) add_n<N>(r.v, r.v, C.p);
    return r;
  }
  Fe operator-() const { return zero() - *this; }
  Fe dbl() const { return *this + *this; }
  // Montgomery product: schoolbook 2N-limb product, then word-by-word reduction (libff mul_reduce shape)
  Fe operator*(const Fe& b) const {
    u64 t[2 * N + 1];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < N; i++) {
      u64 c = 0;
      for (int j = 0; j < N; j++) {
        u128 x = (u128)v[i] * b.v[j] + t[i + j] + c;
        t[i + j] = (u64)x;
        c = (u64)(x >> 64);
      }
      t[i + N] = c;
    }
    for (int i = 0; i < N; i++) {
      u64 m = t[i] * C.inv;
      u64 c = 0;
      for (int j = 0; j < N; j++) {
        u128 x = (u128)m * C.p[j] + t[i + j] + c;
        t[i + j] = (u64)x;
        c = (u64)(x >> 64);
      }
      for (int k = i + N; c && k <= 2 * N; k++) {
        u128 x = (u128)t[k] + c;
        t[k] = (u64)x;
        c = (u64)(x >> 64);
      }
    }
    Fe r;
    memcpy(r.v, t + N, sizeof(r.v));
    if (t[2 * N] || ge<N>(r.v, C.p)) sub_n<N>(r.v, r.v, C.p);
    return r;
  }
  Fe sqr() const { return *this * *this; }
  static Fe from_plain(const u64* x) {
    Fe a, r2;
    memcpy(a.v, x, sizeof(a.v));
    memcpy(r2.v, C.r2, sizeof(r2.v));
    return a * r2;
  }
  void to_plain(u64* out) const {
    Fe o = zero();
    o.v[0] = 1;
    Fe r = *this * o;
    memcpy(out, r.v, sizeof(r.v));
  }
  static Fe from_u64(u64 x) {
    u64 t[N];
    memset(t, 0, sizeof(t));
    t[0] = x;
    return from_plain(t);
  }
  Fe pow(const u64* e, int n) const {
    Fe r = one();
    for (int i = n - 1; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        r = r.sqr();
        if ((e[i] >> b) & 1) r = r * *this;
      }
    return r;
  }
  Fe pow_u64(u64 e) const { return pow(&e, 1); }
  Fe inverse() const {
    u64 e[N], two[N];
    memset(two, 0, sizeof(two));
    two[0] = 2;
    sub_n<N>(e, C.p, two);
    return pow(e, N);
  }
};

template <class F, int NEG_NR>
struct Fe2 {
  static constexpr int LIMBS = 2 * F::LIMBS;
  F c0, c1;
  static Fe2 zero() { return Fe2{F::zero(), F::zero()}; }
  static Fe2 one() { return Fe2{F::one(), F::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const Fe2& b) const { return c0 == b.c0 && c1 == b.c1; }
  bool operator!=(const Fe2& b) const { return !(*this == b); }
  Fe2 operator+(const Fe2& b) const { return Fe2{c0 + b.c0, c1 + b.c1}; }
  Fe2 operator-(const Fe2& b) const { return Fe2{c0 - b.c0, c1 - b.c1}; }
  Fe2 operator-() const { return Fe2{-c0, -c1}; }
  Fe2 dbl() const { return Fe2{c0.dbl(), c1.dbl()}; }
  static F nr(const F& x) {  // x * NEG_NR
    F r = x;
    for (int i = 1; i < NEG_NR; i++) r = r + x;
    return r;
  }
  // schoolbook (4 base multiplications), u^2 = -NEG_NR
  Fe2 operator*(const Fe2& b) const {
    F a0b0 = c0 * b.c0, a1b1 = c1 * b.c1, a0b1 = c0 * b.c1, a1b0 = c1 * b.c0;
    return Fe2{a0b0 - nr(a1b1), a0b1 + a1b0};
  }
  Fe2 sqr() const { return *this * *this; }
  Fe2 inverse() const {
    F n = c0.sqr() + nr(c1.sqr());
    F ni = n.inverse();
    return Fe2{c0 * ni, -(c1 * ni)};
  }
  static Fe2 from_plain(const u64* x) { return Fe2{F::from_plain(x), F::from_plain(x + F::LIMBS)}; }
  void to_plain(u64* out) const {
    c0.to_plain(out);
    c1.to_plain(out + F::LIMBS);
  }
};

// ------------------------------------------------------------------------------------------------
// Jacobian group y^2 = x^3 + b (a = 0); identity has Z == 0
// ------------------------------------------------------------------------------------------------
template <class F>
struct Jac {
  F X, Y, Z;
  static Jac zero() { return Jac{F::zero(), F::one(), F::zero()}; }
  bool is_zero() const { return Z.is_zero(); }
  Jac dbl() const {  // dbl-2009-l
    if (is_zero()) return *this;
    F A = X.sqr(), B = Y.sqr(), C = B.sqr();
    F D = ((X + B).sqr() - A - C).dbl();
    F E = A + A + A;
    F Fv = E.sqr();
    Jac r;
    r.X = Fv - D.dbl();
    r.Y = E * (D - r.X) - C.dbl().dbl().dbl();
    r.Z = (Y * Z).dbl();
    return r;
  }
  Jac add(const Jac& o) const {  // add-2007-bl
    if (is_zero()) return o;
    if (o.is_zero()) return *this;
    F Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
    F U1 = X * Z2Z2, U2 = o.X * Z1Z1;
    F S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
    if (U1 == U2) {
      if (S1 == S2) return dbl();
      return zero();
    }
    F H = U2 - U1;
    F I = H.dbl().sqr();
    F J = H * I;
    F r = (S2 - S1).dbl();
    F V = U1 * I;
    Jac R;
    R.X = r.sqr() - J - V.dbl();
    R.Y = r * (V - R.X) - (S1 * J).dbl();
    R.Z = ((Z + o.Z).sqr() - Z1Z1 - Z2Z2) * H;
    return R;
  }
  // o has Z == 1 (libff "special" form)
  Jac mixed_add(const F& ox, const F& oy) const {  // madd-2007-bl
    if (is_zero()) return Jac{ox, oy, F::one()};
    F Z1Z1 = Z.sqr();
    F U2 = ox * Z1Z1;
    F S2 = oy * Z * Z1Z1;
    if (X == U2) {
      if (Y == S2) return dbl();
      return zero();
    }
    F H = U2 - X;
    F HH = H.sqr();
    F I = HH.dbl().dbl();
    F J = H * I;
    F r = (S2 - Y).dbl();
    F V = X * I;
    Jac R;
    R.X = r.sqr() - J - V.dbl();
    R.Y = r * (V - R.X) - (Y * J).dbl();
    R.Z = (Z + H).sqr() - Z1Z1 - HH;
    return R;
  }
  Jac neg() const { return Jac{X, -Y, Z}; }
  void to_affine(F& x, F& y, bool& inf) const {
    if (is_zero()) {
      inf = true;
      x = F::zero();
      y = F::zero();
      return;
    }
    inf = false;
    F zi = Z.inverse();
    F zi2 = zi.sqr();
    x = X * zi2;
    y = Y * zi2 * zi;
  }
  Jac mul_bits(const u64* k, int nlimbs) const {
    Jac r = zero();
    for (int i = nlimbs - 1; i >= 0; i--)
      for (int b = 63; b >= 0; b--) {
        r = r.dbl();
        if ((k[i] >> b) & 1) r = r.add(*this);
      }
    return r;
  }
};

template <class F>
struct AffPt {
  F x, y;
  bool inf;
};

// ------------------------------------------------------------------------------------------------
// multi-exponentiation (libff multiexp.tcc restated)
// ------------------------------------------------------------------------------------------------
static inline int bigint_num_bits(const u64* e, int n) {
  for (int i = n - 1; i >= 0; i--)
    if (e[i]) return 64 * i + 64 - __builtin_clzll(e[i]);
  return 0;
}
static inline u64 get_bits(const u64* e, int n, int lo, int c) {
  u64 r = 0;
  for (int b = 0; b < c; b++) {
    int bit = lo + b;
    if (bit < 64 * n && ((e[bit / 64] >> (bit % 64)) & 1)) r |= (u64)1 << b;
  }
  return r;
}
static inline int log2_ceil(size_t n) {
  int r = (n & (n - 1)) == 0 ? 0 : 1;
  while (n > 1) {
    n >>= 1;
    r++;
  }
  return r;
}

// multi_exp_inner<BDLO12>: bucket method with unsigned c-bit windows, c = log2(len) - (log2(len)/3 - 2)
template <class F>
static Jac<F> multi_exp_inner_bdlo12(const AffPt<F>* g, const u64* e /* 4 limbs each, plain */, size_t len) {
  if (len == 0) return Jac<F>::zero();
  size_t log2_length = log2_ceil(len);
  if (log2_length < 1) log2_length = 1;
  size_t c = log2_length - (log2_length / 3 - 2);
  if ((long)c < 1) c = 1;
  int num_bits = 0;
  for (size_t i = 0; i < len; i++) {
    int b = bigint_num_bits(e + 4 * i, 4);
    if (b > num_bits) num_bits = b;
  }
  size_t num_groups = (num_bits + c - 1) / c;
  Jac<F> result = Jac<F>::zero();
  bool result_nonzero = false;
  std::vector<Jac<F>> buckets((size_t)1 << c);
  std::vector<char> bucket_nonzero((size_t)1 << c);
  for (size_t k = num_groups; k-- > 0;) {
    if (result_nonzero)
      for (size_t i = 0; i < c; i++) result = result.dbl();
    std::fill(bucket_nonzero.begin(), bucket_nonzero.end(), 0);
    for (size_t i = 0; i < len; i++) {
      if (g[i].inf) continue;
      size_t id = (size_t)get_bits(e + 4 * i, 4, (int)(k * c), (int)c);
      if (id == 0) continue;
      if (bucket_nonzero[id]) {
        buckets[id] = buckets[id].mixed_add(g[i].x, g[i].y);
      } else {
        buckets[id] = Jac<F>{g[i].x, g[i].y, F::one()};
        bucket_nonzero[id] = 1;
      }
    }
    Jac<F> running = Jac<F>::zero();
    bool running_nonzero = false;
    for (size_t i = ((size_t)1 << c) - 1; i > 0; i--) {
      if (bucket_nonzero[i]) {
        running = running_nonzero ? running.add(buckets[i]) : buckets[i];
        running_nonzero = true;
      }
      if (running_nonzero) {
        result = result_nonzero ? result.add(running) : running;
        result_nonzero = true;
      }
    }
  }
  return result;
}

// multi_exp(vec, scalars, chunks): point-range chunks run under OpenMP, partial results added
template <class F>
static Jac<F> multi_exp_chunks(const AffPt<F>* g, const u64* e, size_t len, int chunks) {
  if (chunks < 1) chunks = 1;
  if (len < (size_t)chunks || chunks == 1) return multi_exp_inner_bdlo12<F>(g, e, len);
  size_t one = len / chunks;
  std::vector<Jac<F>> part(chunks);
#pragma omp parallel for schedule(static, 1)
  for (int i = 0; i < chunks; i++) {
    size_t lo = i * one, hi = (i == chunks - 1) ? len : lo + one;
    part[i] = multi_exp_inner_bdlo12<F>(g + lo, e + 4 * lo, hi - lo);
  }
  Jac<F> r = Jac<F>::zero();
  for (int i = 0; i < chunks; i++) r = r.add(part[i]);
  return r;
}

// multi_exp_with_mixed_addition / multi_exp_filter_one_zero: skip 0, add bases with scalar 1 directly,
// bucket the rest
template <class F>
static Jac<F> multi_exp_filtered(const AffPt<F>* g, const u64* e, size_t len, int chunks) {
  std::vector<AffPt<F>> pg;
  std::vector<u64> pe;
  Jac<F> acc = Jac<F>::zero();
  for (size_t i = 0; i < len; i++) {
    const u64* s = e + 4 * i;
    if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
    if (s[0] == 1 && (s[1] | s[2] | s[3]) == 0) {
      if (!g[i].inf) acc = acc.mixed_add(g[i].x, g[i].y);
    } else {
      pg.push_back(g[i]);
      pe.insert(pe.end(), s, s + 4);
    }
  }
  return acc.add(multi_exp_chunks<F>(pg.data(), pe.data(), pg.size(), chunks));
}

template <class F>
static Jac<F> multi_exp_naive(const AffPt<F>* g, const u64* e, size_t len) {
  Jac<F> r = Jac<F>::zero();
  for (size_t i = 0; i < len; i++) {
    if (g[i].inf) continue;
    Jac<F> p{g[i].x, g[i].y, F::one()};
    r = r.add(p.mul_bits(e + 4 * i, 4));
  }
  return r;
}

// ------------------------------------------------------------------------------------------------
// radix-2 evaluation domain (libfqfft basic_radix2_domain restated)
// ------------------------------------------------------------------------------------------------
static inline size_t bitreverse(size_t n, int l) {
  size_t r = 0;
  for (int k = 0; k < l; k++) {
    r = (r << 1) | (n & 1);
    n >>= 1;
  }
  return r;
}

template <class Fr>
static void radix2_fft(std::vector<Fr>& a, const Fr& omega) {
  const size_t n = a.size();
  const int logn = log2_ceil(n);
  for (size_t k = 0; k < n; k++) {
    size_t rk = bitreverse(k, logn);
    if (k < rk) std::swap(a[k], a[rk]);
  }
  size_t m = 1;
  for (int s = 1; s <= logn; s++) {
    u64 ex = n / (2 * m);
    const Fr w_m = omega.pow_u64(ex);
    // the butterflies of one stage are independent: OpenMP over blocks (or over j when blocks are few)
    if (n / (2 * m) >= 64) {
#pragma omp parallel for schedule(static)
      for (size_t k = 0; k < n; k += 2 * m) {
        Fr w = Fr::one();
        for (size_t j = 0; j < m; j++) {
          const Fr t = w * a[k + j + m];
          a[k + j + m] = a[k + j] - t;
          a[k + j] = a[k + j] + t;
          w = w * w_m;
        }
      }
    } else {
      // few large blocks: precompute this stage's twiddles, parallelise inside the block
      std::vector<Fr> tw(m);
      tw[0] = Fr::one();
      for (size_t j = 1; j < m; j++) tw[j] = tw[j - 1] * w_m;
      for (size_t k = 0; k < n; k += 2 * m) {
#pragma omp parallel for schedule(static)
        for (size_t j = 0; j < m; j++) {
          const Fr t = tw[j] * a[k + j + m];
          a[k + j + m] = a[k + j] - t;
          a[k + j] = a[k + j] + t;
        }
      }
    }
    m *= 2;
  }
}

template <class Fr>
struct Domain {
  size_t m;
  Fr omega, omega_inv, m_inv;
  Domain(size_t m_, const Fr& root, int two_adicity) : m(m_) {
    int logm = log2_ceil(m);
    omega = root;
    for (int i = logm; i < two_adicity; i++) omega = omega.sqr();
    omega_inv = omega.inverse();
    m_inv = Fr::from_u64(m).inverse();
  }
  void FFT(std::vector<Fr>& a) const { radix2_fft(a, omega); }
  void iFFT(std::vector<Fr>& a) const {
    radix2_fft(a, omega_inv);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) a[i] = a[i] * m_inv;
  }
  static void multiply_by_coset(std::vector<Fr>& a, const Fr& g) {
    Fr u = g;
    for (size_t i = 1; i < a.size(); i++) {
      a[i] = a[i] * u;
      u = u * g;
    }
  }
  void cosetFFT(std::vector<Fr>& a, const Fr& g) const {
    multiply_by_coset(a, g);
    FFT(a);
  }
  void icosetFFT(std::vector<Fr>& a, const Fr& g) const {
    iFFT(a);
    multiply_by_coset(a, g.inverse());
  }
  Fr compute_vanishing_polynomial(const Fr& t) const { return t.pow_u64(m) - Fr::one(); }
  void divide_by_Z_on_coset(std::vector<Fr>& a, const Fr& g) const {
    Fr zi = compute_vanishing_polynomial(g).inverse();
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) a[i] = a[i] * zi;
  }
};

// ------------------------------------------------------------------------------------------------
// curve bundles
// ------------------------------------------------------------------------------------------------
FieldCtx<4> BN_Q, BN_R, BLS_R;
FieldCtx<6> BLS_Q;
typedef Fe<4, BN_Q> BnFq;
typedef Fe<4, BN_R> BnFr;
typedef Fe<6, BLS_Q> BlsFq;
typedef Fe<4, BLS_R> BlsFr;
typedef Fe2<BnFq, 1> BnFq2;
typedef Fe2<BlsFq, 5> BlsFq2;

static const u64 BN_Q_P[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const u64 BN_R_P[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const u64 BLS_Q_P[6] = {0x8508c00000000001ull, 0x170b5d4430000000ull, 0x1ef3622fba094800ull,
                               0x1a22d9f300f5138full, 0xc63b05c06ca1493bull, 0x01ae3a4617c510eaull};
static const u64 BLS_R_P[4] = {0x0a11800000000001ull, 0x59aa76fed0000001ull, 0x60b44d1e5c37b001ull, 0x12ab655e9a2ca556ull};

struct Init {
  Init() {
    field_init<4>(BN_Q, BN_Q_P);
    field_init<4>(BN_R, BN_R_P);
    field_init<6>(BLS_Q, BLS_Q_P);
    field_init<4>(BLS_R, BLS_R_P);
  }
};
static Init g_init;

// Fr two-adic data: multiplicative generator g, two-adicity s; root_of_unity = g^((r-1)/2^s)
template <class Fr>
struct FrInfo;
template <>
struct FrInfo<BnFr> {
  static constexpr u64 GEN = 5;
  static constexpr int S = 28;
  static const u64* modulus() { return BN_R_P; }
};
template <>
struct FrInfo<BlsFr> {
  static constexpr u64 GEN = 22;
  static constexpr int S = 47;
  static const u64* modulus() { return BLS_R_P; }
};
template <class Fr>
static Fr fr_root_of_unity() {
  u64 e[4], one[4] = {1, 0, 0, 0};
  sub_n<4>(e, FrInfo<Fr>::modulus(), one);
  // e = (r - 1) >> S
  for (int k = 0; k < FrInfo<Fr>::S; k++) {
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 1) | (i + 1 < 4 ? (e[i + 1] << 63) : 0);
  }
  return Fr::from_u64(FrInfo<Fr>::GEN).pow(e, 4);
}

// ------------------------------------------------------------------------------------------------
// C interface (ctypes).  All field elements cross the boundary as plain little-endian 64-bit limbs
// (byte-identical to the product's 32-bit little-endian limbs); the identity is X = Y = 0.
// ------------------------------------------------------------------------------------------------
template <class F>
static void load_points(const u64* in, size_t n, std::vector<AffPt<F>>& out) {
  out.resize(n);
  const int L = F::LIMBS;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) {
    const u64* p = in + 2 * L * i;
    u64 o = 0;
    for (int k = 0; k < 2 * L; k++) o |= p[k];
    out[i].inf = (o == 0);
    out[i].x = F::from_plain(p);
    out[i].y = F::from_plain(p + L);
  }
}
template <class F>
static void store_point(const Jac<F>& r, u64* out) {
  F x, y;
  bool inf;
  r.to_affine(x, y, inf);
  const int L = F::LIMBS;
  if (inf) {
    memset(out, 0, 2 * L * 8);
    return;
  }
  x.to_plain(out);
  y.to_plain(out + L);
}

template <class F>
static int msm_entry(const void* bases, const void* scalars, size_t n, int method, int chunks, void* out,
                     double* seconds) {
  std::vector<AffPt<F>> g;
  load_points<F>((const u64*)bases, n, g);
  const u64* e = (const u64*)scalars;
  auto t0 = std::chrono::steady_clock::now();
  Jac<F> r;
  if (method == 0)
    r = multi_exp_chunks<F>(g.data(), e, n, chunks);
  else if (method == 1)
    r = multi_exp_filtered<F>(g.data(), e, n, chunks);
  else
    r = multi_exp_naive<F>(g.data(), e, n);
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  store_point<F>(r, (u64*)out);
  return 0;
}

template <class Fr>
static int fft_entry(void* data, size_t n, int inverse, int coset, double* seconds) {
  if (n == 0 || (n & (n - 1))) return -1;
  if (log2_ceil(n) > FrInfo<Fr>::S) return -1;
  std::vector<Fr> a(n);
  u64* d = (u64*)data;
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) a[i] = Fr::from_plain(d + 4 * i);
  Domain<Fr> dom(n, fr_root_of_unity<Fr>(), FrInfo<Fr>::S);
  Fr g = Fr::from_u64(FrInfo<Fr>::GEN);
  auto t0 = std::chrono::steady_clock::now();
  if (!inverse && !coset) dom.FFT(a);
  if (inverse && !coset) dom.iFFT(a);
  if (!inverse && coset) dom.cosetFFT(a, g);
  if (inverse && coset) dom.icosetFFT(a, g);
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < n; i++) a[i].to_plain(d + 4 * i);
  return 0;
}

// R1CS in CSR form: for each of the three matrices, row_ptr[nc+1], col[nnz] (variable index, 0 = constant one),
// val[nnz] (Fr plain, 4 limbs)
struct CsrView {
  const u64* row_ptr;
  const uint32_t* col;
  const u64* val;
};

template <class Fr>
static void witness_map(size_t num_inputs, size_t num_vars, size_t nc, const CsrView* abc, const u64* full_plain,
                        size_t m, std::vector<Fr>& H /* m + 1 */) {
  std::vector<Fr> w(num_vars + 1);
  w[0] = Fr::one();
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < num_vars; i++) w[i + 1] = Fr::from_plain(full_plain + 4 * i);
  std::vector<Fr> a[3];
  for (int k = 0; k < 3; k++) {
    a[k].assign(m, Fr::zero());
    std::vector<Fr>& ak = a[k];
    const CsrView& M = abc[k];
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < nc; i++) {
      Fr s = Fr::zero();
      for (u64 t = M.row_ptr[i]; t < M.row_ptr[i + 1]; t++) s = s + Fr::from_plain(M.val + 4 * t) * w[M.col[t]];
      ak[i] = s;
    }
  }
  // input-consistency rows of A (r1cs_to_qap: rows nc .. nc + num_inputs carry the variables 0..num_inputs)
  for (size_t i = 0; i <= num_inputs; i++) a[0][nc + i] = w[i];
  Domain<Fr> dom(m, fr_root_of_unity<Fr>(), FrInfo<Fr>::S);
  Fr g = Fr::from_u64(FrInfo<Fr>::GEN);
  dom.iFFT(a[0]);
  dom.iFFT(a[1]);
  dom.cosetFFT(a[0], g);
  dom.cosetFFT(a[1], g);
  dom.iFFT(a[2]);
  dom.cosetFFT(a[2], g);
  H.assign(m + 1, Fr::zero());
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < m; i++) H[i] = a[0][i] * a[1][i] - a[2][i];
  H.resize(m);
  dom.divide_by_Z_on_coset(H, g);
  dom.icosetFFT(H, g);
  H.push_back(Fr::zero());
}

// proving key view: affine plain points; B query is sparse (indices into 1 || w)
struct PkView {
  const void *alpha_g1, *beta_g1, *beta_g2, *delta_g1, *delta_g2;
  const void* A_query;
  size_t A_len;
  const uint32_t* B_idx;
  const void* B_g2;
  const void* B_g1;
  size_t B_len;
  const void* H_query;
  size_t H_len;
  const void* L_query;
  size_t L_len;
};

template <class Fq, class Fq2T, class Fr>
static int prove_entry(const PkView* pk, size_t num_inputs, size_t num_vars, size_t nc, const CsrView* abc,
                       const u64* full_plain, const u64* r_plain, const u64* s_plain, int chunks, void* proof_out,
                       double* seconds /* [3]: witness map, MSMs, total */) {
  auto t0 = std::chrono::steady_clock::now();
  size_t m = 2;
  while (m < nc + num_inputs + 1) m *= 2;
  std::vector<Fr> H;
  witness_map<Fr>(num_inputs, num_vars, nc, abc, full_plain, m, H);
  auto t1 = std::chrono::steady_clock::now();
  // scalars in plain form: (1 || w) and H coefficients
  std::vector<u64> w1(4 * (num_vars + 1), 0);
  w1[0] = 1;
  memcpy(w1.data() + 4, full_plain, 32 * num_vars);
  std::vector<u64> hs(4 * (m - 1));
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < m - 1; i++) H[i].to_plain(hs.data() + 4 * i);
  std::vector<AffPt<Fq>> A, Bg1, Hq, Lq, one1(1);
  std::vector<AffPt<Fq2T>> Bg2, one2(1);
  load_points<Fq>((const u64*)pk->A_query, pk->A_len, A);
  load_points<Fq>((const u64*)pk->B_g1, pk->B_len, Bg1);
  load_points<Fq2T>((const u64*)pk->B_g2, pk->B_len, Bg2);
  load_points<Fq>((const u64*)pk->H_query, pk->H_len, Hq);
  load_points<Fq>((const u64*)pk->L_query, pk->L_len, Lq);
  if (pk->A_len != num_vars + 1 || pk->H_len != m - 1 || pk->L_len != num_vars - num_inputs) return -1;
  auto t2 = std::chrono::steady_clock::now();
  Jac<Fq> At = multi_exp_filtered<Fq>(A.data(), w1.data(), num_vars + 1, chunks);
  std::vector<u64> bs(4 * pk->B_len);
  for (size_t i = 0; i < pk->B_len; i++) memcpy(bs.data() + 4 * i, w1.data() + 4 * pk->B_idx[i], 32);
  Jac<Fq2T> Bt2 = multi_exp_filtered<Fq2T>(Bg2.data(), bs.data(), pk->B_len, chunks);
  Jac<Fq> Bt1 = multi_exp_filtered<Fq>(Bg1.data(), bs.data(), pk->B_len, chunks);
  Jac<Fq> Ht = multi_exp_chunks<Fq>(Hq.data(), hs.data(), m - 1, chunks);
  Jac<Fq> Lt = multi_exp_filtered<Fq>(Lq.data(), w1.data() + 4 * (num_inputs + 1), num_vars - num_inputs, chunks);
  auto t3 = std::chrono::steady_clock::now();
  auto ld1 = [](const void* p) {
    std::vector<AffPt<Fq>> v;
    load_points<Fq>((const u64*)p, 1, v);
    return v[0].inf ? Jac<Fq>::zero() : Jac<Fq>{v[0].x, v[0].y, Fq::one()};
  };
  auto ld2 = [](const void* p) {
    std::vector<AffPt<Fq2T>> v;
    load_points<Fq2T>((const u64*)p, 1, v);
    return v[0].inf ? Jac<Fq2T>::zero() : Jac<Fq2T>{v[0].x, v[0].y, Fq2T::one()};
  };
  Jac<Fq> alpha1 = ld1(pk->alpha_g1), beta1 = ld1(pk->beta_g1), delta1 = ld1(pk->delta_g1);
  Jac<Fq2T> beta2 = ld2(pk->beta_g2), delta2 = ld2(pk->delta_g2);
  // g_A = alpha + At + r delta ; g1_B = beta + Bt1 + s delta ; g_B = beta2 + Bt2 + s delta2
  Jac<Fq> gA = alpha1.add(At).add(delta1.mul_bits(r_plain, 4));
  Jac<Fq> g1B = beta1.add(Bt1).add(delta1.mul_bits(s_plain, 4));
  Jac<Fq2T> gB = beta2.add(Bt2).add(delta2.mul_bits(s_plain, 4));
  // g_C = Ht + Lt + s g_A + r g1_B - (r s) delta
  Fr rs = Fr::from_plain(r_plain) * Fr::from_plain(s_plain);
  u64 rs_plain[4];
  rs.to_plain(rs_plain);
  Jac<Fq> gC = Ht.add(Lt).add(gA.mul_bits(s_plain, 4)).add(g1B.mul_bits(r_plain, 4)).add(delta1.mul_bits(rs_plain, 4).neg());
  u64* out = (u64*)proof_out;
  store_point<Fq>(gA, out);
  store_point<Fq2T>(gB, out + 2 * Fq::LIMBS);
  store_point<Fq>(gC, out + 2 * Fq::LIMBS + 2 * Fq2T::LIMBS);
  auto t4 = std::chrono::steady_clock::now();
  if (seconds) {
    seconds[0] = std::chrono::duration<double>(t1 - t0).count();
    seconds[1] = std::chrono::duration<double>(t3 - t2).count();
    seconds[2] = std::chrono::duration<double>(t4 - t0).count() - std::chrono::duration<double>(t2 - t1).count();
  }
  return 0;
}

extern "C" {

int oz_max_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void oz_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

// method: 0 = multi_exp<BDLO12> with `chunks` point-range chunks, 1 = multi_exp_with_mixed_addition
// (0/1 filter), 2 = naive sum of double-and-add products
int oz_msm(int curve, int group, const void* bases, const void* scalars, size_t n, int method, int chunks, void* out,
           double* seconds) {
  if (curve == 0 && group == 1) return msm_entry<BnFq>(bases, scalars, n, method, chunks, out, seconds);
  if (curve == 0 && group == 2) return msm_entry<BnFq2>(bases, scalars, n, method, chunks, out, seconds);
  if (curve == 1 && group == 1) return msm_entry<BlsFq>(bases, scalars, n, method, chunks, out, seconds);
  if (curve == 1 && group == 2) return msm_entry<BlsFq2>(bases, scalars, n, method, chunks, out, seconds);
  return -1;
}

// in place on n plain Fr elements: (inverse, coset) selects FFT / iFFT / cosetFFT / icosetFFT with the
// multiplicative generator as the coset shift (the r1cs_to_qap_witness_map usage)
int oz_fft(int curve, void* data, size_t n, int inverse, int coset, double* seconds) {
  if (curve == 0) return fft_entry<BnFr>(data, n, inverse, coset, seconds);
  if (curve == 1) return fft_entry<BlsFr>(data, n, inverse, coset, seconds);
  return -1;
}

// coefficients of H: out has m + 1 plain Fr elements, m = next pow2 >= nc + num_inputs + 1 (force_pow_2_domain)
int oz_witness_map(int curve, size_t num_inputs, size_t num_vars, size_t nc, const CsrView* abc, const void* full,
                   void* out, size_t m) {
  if (curve == 0) {
    std::vector<BnFr> H;
    witness_map<BnFr>(num_inputs, num_vars, nc, abc, (const u64*)full, m, H);
    for (size_t i = 0; i <= m; i++) H[i].to_plain((u64*)out + 4 * i);
    return 0;
  }
  if (curve == 1) {
    std::vector<BlsFr> H;
    witness_map<BlsFr>(num_inputs, num_vars, nc, abc, (const u64*)full, m, H);
    for (size_t i = 0; i <= m; i++) H[i].to_plain((u64*)out + 4 * i);
    return 0;
  }
  return -1;
}

int oz_prove(int curve, const PkView* pk, size_t num_inputs, size_t num_vars, size_t nc, const CsrView* abc,
             const void* full, const void* r, const void* s, int chunks, void* proof_out, double* seconds) {
  if (curve == 0)
    return prove_entry<BnFq, BnFq2, BnFr>(pk, num_inputs, num_vars, nc, abc, (const u64*)full, (const u64*)r,
                                          (const u64*)s, chunks, proof_out, seconds);
  if (curve == 1)
    return prove_entry<BlsFq, BlsFq2, BlsFr>(pk, num_inputs, num_vars, nc, abc, (const u64*)full, (const u64*)r,
                                             (const u64*)s, chunks, proof_out, seconds);
  return -1;
}

}  // extern "C"
