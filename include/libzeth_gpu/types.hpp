// Host-side value types of the libzeth/gpu module: plain-data mirrors of the libff / libsnark types that cross
// Zeth's proving boundary (libzeth/snarks/groth16/groth16_snark.hpp:18-21).  libsnark/libff are not available to
// this build (depends/libsnark is an empty submodule), so the adapter carries its own types with the same member
// names; a Zeth build that has libsnark converts with the few lines shown in INTEGRATION.md.
//
// Every field element is held as a PLAIN integer in little-endian 64-bit limbs (libff::bigint<n>::data order), every
// point as affine X, Y with the identity encoded as X = Y = 0 -- exactly the ZG_FORM_PLAIN layout of zeth_gpu.h,
// so vectors of these types are handed to the C ABI without conversion.
#ifndef LIBZETH_GPU_TYPES_HPP
#define LIBZETH_GPU_TYPES_HPP

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../zeth_gpu.h"

namespace libzeth_gpu
{

/// Error convention of the reference: std::invalid_argument (groth16_snark.tcc:83,98,126,146),
/// std::length_error (r1cs_variable_assignment_serialization.tcc:34-37), anything else std::runtime_error.
inline void check(int status)
{
    if (status == ZG_OK) {
        return;
    }
    if (status == ZG_ERR_INVALID_ARGUMENT) {
        throw std::invalid_argument(std::string("zeth_gpu: ") + zg_strerror(status));
    }
    if (status == ZG_ERR_LENGTH) {
        throw std::length_error(std::string("zeth_gpu: ") + zg_strerror(status));
    }
    throw std::runtime_error(std::string("zeth_gpu: ") + zg_strerror(status));
}

/// n 64-bit limbs, little-endian, plain (non-Montgomery) value.
template<size_t n> struct field_element {
    static constexpr size_t num_limbs = n;
    uint64_t data[n];

    static field_element zero()
    {
        field_element r;
        memset(r.data, 0, sizeof(r.data));
        return r;
    }
    static field_element one() { return field_element(1); }
    field_element() { memset(data, 0, sizeof(data)); }
    explicit field_element(uint64_t v)
    {
        memset(data, 0, sizeof(data));
        data[0] = v;
    }
    bool is_zero() const
    {
        uint64_t o = 0;
        for (size_t i = 0; i < n; i++) {
            o |= data[i];
        }
        return o == 0;
    }
    bool operator==(const field_element &o) const { return memcmp(data, o.data, sizeof(data)) == 0; }
    bool operator!=(const field_element &o) const { return !(*this == o); }
    /// numeric comparison (used to range-check values read from files)
    bool less_than(const uint64_t *modulus) const
    {
        for (size_t i = n; i-- > 0;) {
            if (data[i] != modulus[i]) {
                return data[i] < modulus[i];
            }
        }
        return false;
    }
};

template<typename BaseT> struct fp2_element {
    using base_type = BaseT;
    BaseT c0, c1;
    static fp2_element zero() { return fp2_element{BaseT::zero(), BaseT::zero()}; }
    static fp2_element one() { return fp2_element{BaseT::one(), BaseT::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const fp2_element &o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const fp2_element &o) const { return !(*this == o); }
};

/// Affine point; the identity is X = Y = 0 (the ABI's encoding, not on either curve since b != 0).
template<typename CoordT, int GroupId> struct group_element {
    using coordinate_type = CoordT;
    static constexpr int group_id = GroupId;
    CoordT X, Y;
    static group_element zero() { return group_element{CoordT::zero(), CoordT::zero()}; }
    bool is_zero() const { return X.is_zero() && Y.is_zero(); }
    bool operator==(const group_element &o) const { return X == o.X && Y == o.Y; }
    bool operator!=(const group_element &o) const { return !(*this == o); }
};

/// Curve tags named after libff's (zeth_tool/tool_common.hpp:44-67 dispatches on the same two).
struct alt_bn128_pp {
    static constexpr int curve_id = ZG_CURVE_ALT_BN128;
    static constexpr const char *name = "alt-bn128";
    using Fr_type = field_element<4>;
    using Fq_type = field_element<4>;
    /// group order r and the libff multiplicative generator of Fr (coset shift of the QAP domain)
    static constexpr uint64_t r[4] = {
        0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    static constexpr uint64_t q[4] = {
        0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    static constexpr unsigned r_bits = 254;
    static constexpr uint64_t multiplicative_generator = 5;
    static constexpr unsigned two_adicity = 28;
    static void init_public_params() {}
};

struct bls12_377_pp {
    static constexpr int curve_id = ZG_CURVE_BLS12_377;
    static constexpr const char *name = "bls12-377";
    using Fr_type = field_element<4>;
    using Fq_type = field_element<6>;
    static constexpr uint64_t r[4] = {
        0x0a11800000000001ull, 0x59aa76fed0000001ull, 0x60b44d1e5c37b001ull, 0x12ab655e9a2ca556ull};
    static constexpr uint64_t q[6] = {0x8508c00000000001ull,
                                      0x170b5d4430000000ull,
                                      0x1ef3622fba094800ull,
                                      0x1a22d9f300f5138full,
                                      0xc63b05c06ca1493bull,
                                      0x01ae3a4617c510eaull};
    static constexpr unsigned r_bits = 253;
    static constexpr uint64_t multiplicative_generator = 22;
    static constexpr unsigned two_adicity = 47;
    static void init_public_params() {}
};

template<typename ppT> using Fr = typename ppT::Fr_type;
template<typename ppT> using Fq = typename ppT::Fq_type;
template<typename ppT> using Fq2 = fp2_element<typename ppT::Fq_type>;
template<typename ppT> using G1 = group_element<Fq<ppT>, ZG_GROUP_G1>;
template<typename ppT> using G2 = group_element<Fq2<ppT>, ZG_GROUP_G2>;
template<typename ppT> using Fr_vector = std::vector<Fr<ppT>>;
template<typename ppT> using G1_vector = std::vector<G1<ppT>>;
template<typename ppT> using G2_vector = std::vector<G2<ppT>>;

// ---- libsnark-shaped containers (member names as in libsnark, which the reference's codecs use:
//      libzeth/serialization/stream_utils.tcc:92-155, r1cs_serialization.tcc:199-275) ----

using var_index_t = size_t;

template<typename FieldT> struct linear_term {
    var_index_t index; // 0 = the constant ONE
    FieldT coeff;
};

template<typename FieldT> struct linear_combination {
    std::vector<linear_term<FieldT>> terms;
    void add_term(const linear_term<FieldT> &t) { terms.push_back(t); }
    void add_term(var_index_t index, uint64_t coeff) { terms.push_back({index, FieldT(coeff)}); }
};

template<typename FieldT> struct r1cs_constraint {
    linear_combination<FieldT> a, b, c;
};

template<typename FieldT> struct r1cs_constraint_system {
    size_t primary_input_size = 0;
    size_t auxiliary_input_size = 0;
    std::vector<r1cs_constraint<FieldT>> constraints;
    size_t num_inputs() const { return primary_input_size; }
    size_t num_variables() const { return primary_input_size + auxiliary_input_size; }
    size_t num_constraints() const { return constraints.size(); }
    void add_constraint(const r1cs_constraint<FieldT> &c) { constraints.push_back(c); }
    /// libsnark's r1cs_constraint_system::swap_AB_if_beneficial() [restated from the published libsnark source, which is
    /// absent from /root/reference; the reference shows its use at libzeth/tests/mpc/groth16/mpc_test.cpp:42,164,340]:
    /// when more variables occur in the B side of the constraints than in the A side, swap a and b in every constraint
    /// (a * b = c is symmetric), so that the B query -- evaluated in G2 AND G1 -- is the sparser one.
    /// Returns whether the sides were swapped.
    bool swap_AB_if_beneficial()
    {
        std::vector<bool> touched_by_A(num_variables() + 1, false), touched_by_B(num_variables() + 1, false);
        for (const r1cs_constraint<FieldT> &c : constraints) {
            for (const linear_term<FieldT> &t : c.a.terms) {
                touched_by_A.at(t.index) = true;
            }
            for (const linear_term<FieldT> &t : c.b.terms) {
                touched_by_B.at(t.index) = true;
            }
        }
        size_t non_zero_A_count = 0, non_zero_B_count = 0;
        for (size_t i = 0; i < num_variables() + 1; ++i) {
            non_zero_A_count += touched_by_A[i] ? 1 : 0;
            non_zero_B_count += touched_by_B[i] ? 1 : 0;
        }
        if (non_zero_B_count <= non_zero_A_count) {
            return false;
        }
        for (r1cs_constraint<FieldT> &c : constraints) {
            std::swap(c.a, c.b);
        }
        return true;
    }
};

template<typename FieldT> using r1cs_primary_input = std::vector<FieldT>;
template<typename FieldT> using r1cs_auxiliary_input = std::vector<FieldT>;
template<typename FieldT> using r1cs_variable_assignment = std::vector<FieldT>;

/// The part of libsnark::protoboard<FieldT> the proving entry point reads (circuit_wrapper.tcc:150-170 fills it, then
/// groth16_snark::generate_proof(pk, pb) takes pb.primary_input() / pb.auxiliary_input(), groth16_snark.tcc:27-34):
/// the full assignment without the constant ONE, primary inputs first.
template<typename FieldT> struct protoboard {
    r1cs_variable_assignment<FieldT> values;
    size_t num_inputs_ = 0;
    void set_input_sizes(size_t primary_input_size)
    {
        if (primary_input_size > values.size()) {
            throw std::invalid_argument("protoboard: more inputs than variables");
        }
        num_inputs_ = primary_input_size;
    }
    size_t num_inputs() const { return num_inputs_; }
    size_t num_variables() const { return values.size(); }
    /// variable `index` (1-based, as in libsnark; 0 is the constant ONE)
    FieldT &val(var_index_t index) { return values.at(index - 1); }
    const r1cs_variable_assignment<FieldT> &full_variable_assignment() const { return values; }
    r1cs_primary_input<FieldT> primary_input() const { return r1cs_primary_input<FieldT>(values.begin(), values.begin() + num_inputs_); }
    r1cs_auxiliary_input<FieldT> auxiliary_input() const { return r1cs_auxiliary_input<FieldT>(values.begin() + num_inputs_, values.end()); }
};

template<typename T> struct sparse_vector {
    std::vector<size_t> indices;
    std::vector<T> values;
    size_t domain_size_ = 0;
};

template<typename T1, typename T2> struct knowledge_commitment {
    T1 g;
    T2 h;
};

template<typename T1, typename T2> using knowledge_commitment_vector = sparse_vector<knowledge_commitment<T1, T2>>;

/// libsnark::accumulation_vector<T>: `first` plus a sparse `rest` (the reference reads both members:
/// groth16_snark.tcc:62-72, 216-224; stream_utils.tcc:125-139)
template<typename T> struct accumulation_vector {
    T first;
    sparse_vector<T> rest;
    accumulation_vector() : first(T::zero()) {}
    /// first = v[0], rest = v[1..] at indices 0..n-2 (libsnark's accumulation_vector(first, std::vector rest))
    explicit accumulation_vector(const std::vector<T> &v) : first(v.empty() ? T::zero() : v[0])
    {
        for (size_t i = 1; i < v.size(); i++) {
            rest.indices.push_back(i - 1);
            rest.values.push_back(v[i]);
        }
        rest.domain_size_ = rest.values.size();
    }
    size_t size() const { return rest.domain_size_ + 1; }
};

/// r1cs_gg_ppzksnark_proving_key<ppT> (fields as enumerated by groth16_snark.tcc:101-110)
template<typename ppT> struct r1cs_gg_ppzksnark_proving_key {
    G1<ppT> alpha_g1;
    G1<ppT> beta_g1;
    G2<ppT> beta_g2;
    G1<ppT> delta_g1;
    G2<ppT> delta_g2;
    G1_vector<ppT> A_query;
    knowledge_commitment_vector<G2<ppT>, G1<ppT>> B_query;
    G1_vector<ppT> H_query;
    G1_vector<ppT> L_query;
    r1cs_constraint_system<Fr<ppT>> constraint_system;
};

/// r1cs_gg_ppzksnark_verification_key<ppT> as this fork's verifier uses it (fields as read by groth16_snark.tcc:62-72,
/// 86-90, 119-123; gamma is the G2 generator: SURVEY KAT-6)
template<typename ppT> struct r1cs_gg_ppzksnark_verification_key {
    G1<ppT> alpha_g1;
    G2<ppT> beta_g2;
    G2<ppT> delta_g2;
    accumulation_vector<G1<ppT>> ABC_g1; // num_inputs + 1 entries
};

/// r1cs_gg_ppzksnark_keypair<ppT> (groth16_snark.hpp:20; groth16_snark.tcc:177-191)
template<typename ppT> struct r1cs_gg_ppzksnark_keypair {
    r1cs_gg_ppzksnark_proving_key<ppT> pk;
    r1cs_gg_ppzksnark_verification_key<ppT> vk;
};

/// r1cs_gg_ppzksnark_proof<ppT> (groth16_snark.tcc:154-156, 163-165)
template<typename ppT> struct r1cs_gg_ppzksnark_proof {
    G1<ppT> g_A;
    G2<ppT> g_B;
    G1<ppT> g_C;
    bool operator==(const r1cs_gg_ppzksnark_proof &o) const { return g_A == o.g_A && g_B == o.g_B && g_C == o.g_C; }
};

} // namespace libzeth_gpu

#endif // LIBZETH_GPU_TYPES_HPP
