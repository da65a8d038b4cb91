// Key generation for the libzeth/gpu module: groth16_snark<ppT>::generate_setup (libzeth/snarks/groth16/groth16_snark.hpp:27,
// groth16_snark.tcc:18-25, which calls libsnark::r1cs_gg_ppzksnark_generator<ppT>(cs, true)) and the from-secrets form
// the reference's MPC test checks the ceremony against (r1cs_gg_ppzksnark_generator_from_secrets(cs, t, alpha, beta,
// delta, G1::one(), G2::one(), true), libzeth/tests/mpc/groth16/mpc_test.cpp:334-366).
//
// Split of the work, as in libsnark's generator:
//   host    r1cs_to_qap_instance_map_with_evaluation at t: the Lagrange basis of the power-of-two domain evaluated at t
//           (one batch inversion), At/Bt/Ct accumulated over the constraints (plus the num_inputs + 1 extra rows of A),
//           the scalar vectors of the four queries -- plain 64-bit-limb Montgomery arithmetic below, a few million
//           field products for a JoinSplit-sized system
//   device  libff::batch_exp: the fixed-base exponentiations s_i * G of every query (zg_bases_batch_exp)
// The key has the structure asserted by mpc_test.cpp:282-331 (tests/test_gpu_host_adapter.py compares it with the
// oracle's generator_from_secrets byte for byte in the proving key's on-disk format).
#ifndef LIBZETH_GPU_GROTH16_GENERATOR_HPP
#define LIBZETH_GPU_GROTH16_GENERATOR_HPP

#include "gpu_groth16_snark.hpp"

namespace libzeth_gpu
{

namespace internal
{

/// Fr on the host: 4 x 64-bit limbs, Montgomery form inside, plain field_element<4> at the edges.
template<typename ppT> class host_fr
{
public:
    using limbs = field_element<4>;

    host_fr()
    {
        // -r^-1 mod 2^64 (Newton), R mod r and R^2 mod r (by doubling)
        uint64_t x = 1;
        for (int i = 0; i < 6; i++) {
            x *= 2 - ppT::r[0] * x;
        }
        inv_ = ~x + 1;
        limbs one_plain(1);
        limbs acc = one_plain;
        for (int i = 0; i < 256; i++) {
            acc = add(acc, acc);
        }
        r1_ = acc;
        for (int i = 0; i < 256; i++) {
            acc = add(acc, acc);
        }
        r2_ = acc;
    }

    limbs to_mont(const limbs &a) const { return mul(a, r2_); }
    limbs from_mont(const limbs &a) const { return mul(a, limbs(1)); }
    limbs one() const { return r1_; }

    static limbs add(const limbs &a, const limbs &b)
    {
        limbs s;
        unsigned __int128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (unsigned __int128)a.data[i] + b.data[i];
            s.data[i] = (uint64_t)c;
            c >>= 64;
        }
        return reduce_once(s, (uint64_t)c);
    }
    static limbs sub(const limbs &a, const limbs &b)
    {
        limbs d;
        uint64_t borrow = 0;
        for (int i = 0; i < 4; i++) {
            const unsigned __int128 t = (unsigned __int128)a.data[i] - b.data[i] - borrow;
            d.data[i] = (uint64_t)t;
            borrow = (uint64_t)(t >> 64) & 1;
        }
        if (borrow) {
            unsigned __int128 c = 0;
            for (int i = 0; i < 4; i++) {
                c += (unsigned __int128)d.data[i] + ppT::r[i];
                d.data[i] = (uint64_t)c;
                c >>= 64;
            }
        }
        return d;
    }
    /// Montgomery product a b / 2^256 mod r (CIOS)
    limbs mul(const limbs &a, const limbs &b) const
    {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            unsigned __int128 c = 0;
            for (int j = 0; j < 4; j++) {
                c += (unsigned __int128)a.data[j] * b.data[i] + t[j];
                t[j] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (uint64_t)c;
            t[5] = (uint64_t)(c >> 64);
            const uint64_t m = t[0] * inv_;
            c = (unsigned __int128)m * ppT::r[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; j++) {
                c += (unsigned __int128)m * ppT::r[j] + t[j];
                t[j - 1] = (uint64_t)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (uint64_t)c;
            t[4] = t[5] + (uint64_t)(c >> 64);
        }
        limbs s;
        memcpy(s.data, t, sizeof(s.data));
        return reduce_once(s, t[4]);
    }
    limbs pow(limbs base, const uint64_t *e, size_t nlimbs) const
    {
        limbs acc = r1_;
        for (size_t i = 0; i < 64 * nlimbs; i++) {
            if ((e[i / 64] >> (i % 64)) & 1) {
                acc = mul(acc, base);
            }
            base = mul(base, base);
        }
        return acc;
    }
    limbs pow(const limbs &base, uint64_t e) const { return pow(base, &e, 1); }
    /// a^(r - 2)
    limbs inverse(const limbs &a) const
    {
        uint64_t e[4];
        memcpy(e, ppT::r, sizeof(e));
        e[0] -= 2; // r is odd and > 2: no borrow
        return pow(a, e, 4);
    }
    /// every element of v replaced by its inverse (all non-zero): one inversion, 3 n products
    void batch_inverse(std::vector<limbs> &v) const
    {
        std::vector<limbs> prefix(v.size());
        limbs acc = r1_;
        for (size_t i = 0; i < v.size(); i++) {
            prefix[i] = acc;
            acc = mul(acc, v[i]);
        }
        limbs inv = inverse(acc);
        for (size_t i = v.size(); i-- > 0;) {
            const limbs vi = v[i];
            v[i] = mul(inv, prefix[i]);
            inv = mul(inv, vi);
        }
    }

private:
    uint64_t inv_;
    limbs r1_, r2_;
    static limbs reduce_once(const limbs &s, uint64_t carry)
    {
        limbs d;
        uint64_t borrow = 0;
        for (int i = 0; i < 4; i++) {
            const unsigned __int128 t = (unsigned __int128)s.data[i] - ppT::r[i] - borrow;
            d.data[i] = (uint64_t)t;
            borrow = (uint64_t)(t >> 64) & 1;
        }
        return (carry || !borrow) ? d : s;
    }
};

/// s_i * G for plain scalars, on the device (libff::batch_exp)
template<typename ppT, typename GroupT> std::vector<GroupT> batch_exp(const std::vector<Fr<ppT>> &scalars)
{
    std::vector<GroupT> out(scalars.size());
    if (scalars.empty()) {
        return out;
    }
    device_context &dc = device_context::get();
    std::lock_guard<std::mutex> lock(dc.mutex());
    zg_bases *b = nullptr;
    check(zg_bases_batch_exp(dc.ctx(), ppT::curve_id, GroupT::group_id, scalars.data(), scalars.size(), &b));
    const int s = zg_bases_read(b, 0, scalars.size(), ZG_FORM_PLAIN, out.data());
    zg_bases_free(b);
    check(s);
    return out;
}

} // namespace internal

/// The evaluation of the QAP of `cs` at t over the power-of-two domain (force_pow_2_domain = true, as Zeth always
/// passes): At, Bt, Ct over the variables 0..num_variables, Zt = t^m - 1.  Plain values.
template<typename ppT> struct qap_instance_evaluation {
    size_t domain_size = 0;
    Fr_vector<ppT> At, Bt, Ct;
    Fr<ppT> Zt;
};

template<typename ppT> qap_instance_evaluation<ppT> r1cs_to_qap_instance_map_with_evaluation(const r1cs_constraint_system<Fr<ppT>> &cs, const Fr<ppT> &t)
{
    using F = Fr<ppT>;
    const internal::host_fr<ppT> fr;
    const size_t nc = cs.num_constraints(), ni = cs.num_inputs(), nv = cs.num_variables();
    size_t m = 2; // libfqfft::basic_radix2_domain needs m > 1
    unsigned log_m = 1;
    while (m < nc + ni + 1) {
        m <<= 1;
        log_m++;
    }
    if (log_m > ppT::two_adicity) {
        throw std::invalid_argument("r1cs_to_qap: constraint system too large for the scalar field's 2-adicity");
    }
    // omega = g^((r - 1) / 2^s) squared down to order m (libff: root_of_unity = multiplicative_generator^t)
    uint64_t e[4];
    memcpy(e, ppT::r, sizeof(e));
    e[0] -= 1;
    for (unsigned i = 0; i < ppT::two_adicity; i++) { // (r - 1) >> s
        for (int k = 0; k < 4; k++) {
            e[k] = (e[k] >> 1) | (k < 3 ? e[k + 1] << 63 : 0);
        }
    }
    F omega = fr.pow(fr.to_mont(F(ppT::multiplicative_generator)), e, 4);
    for (unsigned i = log_m; i < ppT::two_adicity; i++) {
        omega = fr.mul(omega, omega);
    }
    const F tm = fr.to_mont(t);
    const F Zm = fr.sub(fr.pow(tm, (uint64_t)m), fr.one());
    if (Zm.is_zero()) {
        throw std::invalid_argument("r1cs_to_qap: t is an element of the evaluation domain");
    }
    // u_i = Z(t) / (m (t - w^i)) * w^i
    std::vector<F> w(m), den(m);
    F acc = fr.one();
    for (size_t i = 0; i < m; i++) {
        w[i] = acc;
        den[i] = fr.sub(tm, acc);
        acc = fr.mul(acc, omega);
    }
    fr.batch_inverse(den);
    const F c = fr.mul(Zm, fr.inverse(fr.to_mont(F((uint64_t)m))));
    std::vector<F> u(m);
    for (size_t i = 0; i < m; i++) {
        u[i] = fr.mul(fr.mul(c, w[i]), den[i]);
    }
    std::vector<F> At(nv + 1), Bt(nv + 1), Ct(nv + 1);
    for (size_t i = 0; i <= ni; i++) {
        At[i] = u[nc + i];
    }
    auto accumulate = [&](std::vector<F> &dst, const linear_combination<F> &lc, const F &ui) {
        for (const auto &term : lc.terms) {
            if (term.index > nv) {
                throw std::invalid_argument("r1cs_to_qap: variable index outside the constraint system");
            }
            dst[term.index] = fr.add(dst[term.index], fr.mul(ui, fr.to_mont(term.coeff)));
        }
    };
    for (size_t i = 0; i < nc; i++) {
        accumulate(At, cs.constraints[i].a, u[i]);
        accumulate(Bt, cs.constraints[i].b, u[i]);
        accumulate(Ct, cs.constraints[i].c, u[i]);
    }
    qap_instance_evaluation<ppT> q;
    q.domain_size = m;
    q.At.resize(nv + 1);
    q.Bt.resize(nv + 1);
    q.Ct.resize(nv + 1);
    for (size_t i = 0; i <= nv; i++) {
        q.At[i] = fr.from_mont(At[i]);
        q.Bt[i] = fr.from_mont(Bt[i]);
        q.Ct[i] = fr.from_mont(Ct[i]);
    }
    q.Zt = fr.from_mont(Zm);
    return q;
}

/// r1cs_gg_ppzksnark_generator_from_secrets(cs, t, alpha, beta, delta, G1::one(), G2::one(), true): all secrets plain
/// values in [0, r); delta must be non-zero, t outside the domain.
template<typename ppT>
r1cs_gg_ppzksnark_keypair<ppT> r1cs_gg_ppzksnark_generator_from_secrets(
    const r1cs_constraint_system<Fr<ppT>> &cs, const Fr<ppT> &t, const Fr<ppT> &alpha, const Fr<ppT> &beta, const Fr<ppT> &delta)
{
    using F = Fr<ppT>;
    for (const F *s : {&t, &alpha, &beta, &delta}) {
        if (!s->less_than(ppT::r)) {
            throw std::invalid_argument("generator: secret not reduced modulo the group order");
        }
    }
    if (delta.is_zero()) {
        throw std::invalid_argument("generator: delta must be invertible");
    }
    const internal::host_fr<ppT> fr;
    const qap_instance_evaluation<ppT> q = r1cs_to_qap_instance_map_with_evaluation<ppT>(cs, t);
    const size_t m = q.domain_size, ni = cs.num_inputs(), nv = cs.num_variables();
    const F tm = fr.to_mont(t), am = fr.to_mont(alpha), bm = fr.to_mont(beta);
    const F dinv = fr.inverse(fr.to_mont(delta));

    // H query: t^i Z(t) / delta, i < m - 1
    Fr_vector<ppT> h_sc(m - 1);
    F acc = fr.mul(fr.to_mont(q.Zt), dinv);
    for (size_t i = 0; i + 1 < m; i++) {
        h_sc[i] = fr.from_mont(acc);
        acc = fr.mul(acc, tm);
    }
    // (beta A_i + alpha B_i + C_i): the first num_inputs + 1 go to the verification key, the rest / delta to L
    Fr_vector<ppT> abc_sc(ni + 1), l_sc(nv - ni), b_sc;
    std::vector<size_t> b_idx;
    for (size_t i = 0; i <= nv; i++) {
        const F abc = fr.add(fr.add(fr.mul(bm, fr.to_mont(q.At[i])), fr.mul(am, fr.to_mont(q.Bt[i]))), fr.to_mont(q.Ct[i]));
        if (i <= ni) {
            abc_sc[i] = fr.from_mont(abc);
        } else {
            l_sc[i - ni - 1] = fr.from_mont(fr.mul(abc, dinv));
        }
        if (!q.Bt[i].is_zero()) { // the B query is sparse: zero entries are not stored
            b_idx.push_back(i);
            b_sc.push_back(q.Bt[i]);
        }
    }

    r1cs_gg_ppzksnark_keypair<ppT> kp;
    const G1_vector<ppT> singles_g1 = internal::batch_exp<ppT, G1<ppT>>({alpha, beta, delta});
    const G2_vector<ppT> singles_g2 = internal::batch_exp<ppT, G2<ppT>>({beta, delta});
    kp.pk.alpha_g1 = singles_g1[0];
    kp.pk.beta_g1 = singles_g1[1];
    kp.pk.delta_g1 = singles_g1[2];
    kp.pk.beta_g2 = singles_g2[0];
    kp.pk.delta_g2 = singles_g2[1];
    kp.pk.A_query = internal::batch_exp<ppT, G1<ppT>>(q.At);
    const G2_vector<ppT> b_g2 = internal::batch_exp<ppT, G2<ppT>>(b_sc);
    const G1_vector<ppT> b_g1 = internal::batch_exp<ppT, G1<ppT>>(b_sc);
    kp.pk.B_query.domain_size_ = nv + 1;
    kp.pk.B_query.indices = b_idx;
    kp.pk.B_query.values.resize(b_idx.size());
    for (size_t k = 0; k < b_idx.size(); k++) {
        kp.pk.B_query.values[k].g = b_g2[k];
        kp.pk.B_query.values[k].h = b_g1[k];
    }
    kp.pk.H_query = internal::batch_exp<ppT, G1<ppT>>(h_sc);
    kp.pk.L_query = internal::batch_exp<ppT, G1<ppT>>(l_sc);
    kp.pk.constraint_system = cs;
    kp.vk.alpha_g1 = kp.pk.alpha_g1;
    kp.vk.beta_g2 = kp.pk.beta_g2;
    kp.vk.delta_g2 = kp.pk.delta_g2;
    kp.vk.ABC_g1 = accumulation_vector<G1<ppT>>(internal::batch_exp<ppT, G1<ppT>>(abc_sc));
    return kp;
}

/// r1cs_gg_ppzksnark_generator<ppT>(cs, true): like libsnark's, it works on a copy of the system with
/// swap_AB_if_beneficial() applied (the from-secrets form above does NOT swap: callers that compare the two apply the
/// swap themselves, mpc_test.cpp:42,164,340), and the key carries that copy.  The secrets are drawn from the OS entropy
/// source (t outside the domain and delta non-zero with overwhelming probability; redrawn otherwise).
template<typename ppT> r1cs_gg_ppzksnark_keypair<ppT> r1cs_gg_ppzksnark_generator(const r1cs_constraint_system<Fr<ppT>> &cs)
{
    r1cs_constraint_system<Fr<ppT>> cs_copy(cs);
    cs_copy.swap_AB_if_beneficial();
    for (;;) {
        const Fr<ppT> t = random_field_element<ppT>(), alpha = random_field_element<ppT>(), beta = random_field_element<ppT>(),
                      delta = random_field_element<ppT>();
        if (delta.is_zero()) {
            continue;
        }
        try {
            return r1cs_gg_ppzksnark_generator_from_secrets<ppT>(cs_copy, t, alpha, beta, delta);
        } catch (const std::invalid_argument &e) {
            if (std::string(e.what()).find("evaluation domain") == std::string::npos) {
                throw;
            }
        }
    }
}

/// groth16_snark<ppT>::generate_setup (groth16_snark.hpp:27; .tcc:18-25 passes pb.get_constraint_system())
template<typename ppT> r1cs_gg_ppzksnark_keypair<ppT> generate_setup(const r1cs_constraint_system<Fr<ppT>> &cs)
{
    return r1cs_gg_ppzksnark_generator<ppT>(cs);
}

} // namespace libzeth_gpu

#endif // LIBZETH_GPU_GROTH16_GENERATOR_HPP
