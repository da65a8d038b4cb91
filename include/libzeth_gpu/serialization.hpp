// Zeth's on-disk encodings for the objects that cross the proving boundary, restated for the
// plain-data types of types.hpp.  Formats (all citations under /root/reference):
//   base field element   plain (non-Montgomery) fixed-size BIG-endian integer   libzeth/core/field_element_utils.hpp:33-37
//   Fq2 element          c1 then c0 (binary and JSON)                            client/test_contracts/test_bls12_377_contract.py:66-75
//   group element        affine X || Y, the identity written as (0, 1)           libzeth/core/group_element_utils.tcc:93-113
//   size_t / var_index_t native 8-byte little-endian                             libzeth/serialization/stream_utils.tcc:23-35
//   collection           size_t n, then n members                                stream_utils.tcc:65-90
//   sparse_vector        size_t domain_size, size_t n, n x (size_t index, value) stream_utils.tcc:92-123
//   knowledge_commitment g then h                                                stream_utils.tcc:141-155
//   linear_combination   uint32 n_terms, n x (var_index_t, Fr)                   serialization/r1cs_serialization.tcc:199-235
//   r1cs                 size_t primary, size_t auxiliary, collection of (a, b, c) r1cs_serialization.tcc:255-275
//   proving key          alpha_g1 beta_g1 beta_g2 delta_g1 delta_g2 A B H L r1cs  snarks/groth16/groth16_snark.tcc:93-148
//   proof                g_A g_B g_C                                             groth16_snark.tcc:159-175
//   assignment           collection of Fr, primary inputs first                  serialization/r1cs_variable_assignment_serialization.tcc:15-77
#ifndef LIBZETH_GPU_SERIALIZATION_HPP
#define LIBZETH_GPU_SERIALIZATION_HPP

#include <fstream>
#include <istream>
#include <limits>
#include <ostream>
#include <sstream>

#include "types.hpp"

namespace libzeth_gpu
{

// ---- fundamental values (stream_utils.tcc:14-35) ----
template<typename T> T read_bytes(std::istream &in_s)
{
    T val;
    in_s.read((char *)(&val), sizeof(T));
    if (!in_s) {
        throw std::ios_base::failure("unexpected end of stream");
    }
    return val;
}

template<typename T> void write_bytes(const T &val, std::ostream &out_s)
{
    out_s.write((const char *)(&val), sizeof(T));
}

/// libtool::open_binary_input_file (libtool/tool_util.cpp:10-17): failures surface as stream exceptions
inline std::ifstream open_binary_input_file(const std::string &filename)
{
    std::ifstream in_s(filename.c_str(), std::ios_base::in | std::ios_base::binary);
    if (!in_s) {
        throw std::ios_base::failure("cannot open " + filename);
    }
    in_s.exceptions(std::ios_base::eofbit | std::ios_base::badbit | std::ios_base::failbit);
    return in_s;
}

inline std::ofstream open_binary_output_file(const std::string &filename)
{
    std::ofstream out_s(filename.c_str(), std::ios_base::out | std::ios_base::binary);
    if (!out_s) {
        throw std::ios_base::failure("cannot open " + filename);
    }
    return out_s;
}

// ---- field elements ----
template<size_t n> void field_element_write_bytes(const field_element<n> &el, std::ostream &out_s)
{
    uint8_t buf[8 * n];
    for (size_t i = 0; i < n; i++) {
        const uint64_t limb = el.data[n - 1 - i];
        for (size_t b = 0; b < 8; b++) {
            buf[8 * i + b] = (uint8_t)(limb >> (56 - 8 * b));
        }
    }
    out_s.write((const char *)buf, sizeof(buf));
}

template<size_t n> void field_element_read_bytes(field_element<n> &el, std::istream &in_s)
{
    uint8_t buf[8 * n];
    in_s.read((char *)buf, sizeof(buf));
    if (!in_s) {
        throw std::ios_base::failure("unexpected end of stream");
    }
    for (size_t i = 0; i < n; i++) {
        uint64_t limb = 0;
        for (size_t b = 0; b < 8; b++) {
            limb = (limb << 8) | buf[8 * i + b];
        }
        el.data[n - 1 - i] = limb;
    }
}

template<typename BaseT> void field_element_write_bytes(const fp2_element<BaseT> &el, std::ostream &out_s)
{
    field_element_write_bytes(el.c1, out_s);
    field_element_write_bytes(el.c0, out_s);
}

template<typename BaseT> void field_element_read_bytes(fp2_element<BaseT> &el, std::istream &in_s)
{
    field_element_read_bytes(el.c1, in_s);
    field_element_read_bytes(el.c0, in_s);
}

/// "0x"-prefixed fixed-width hex, the JSON form of libff::field_write<encoding_json, form_plain>
template<size_t n> std::string base_field_element_to_hex(const field_element<n> &el)
{
    static const char digits[] = "0123456789abcdef";
    std::string s;
    s.reserve(16 * n);
    for (size_t i = n; i-- > 0;) {
        for (int b = 60; b >= 0; b -= 4) {
            s.push_back(digits[(el.data[i] >> b) & 0xf]);
        }
    }
    return s;
}

template<size_t n> void field_element_write_json(const field_element<n> &el, std::ostream &out_s)
{
    out_s << "\"0x" << base_field_element_to_hex(el) << "\"";
}

template<typename BaseT> void field_element_write_json(const fp2_element<BaseT> &el, std::ostream &out_s)
{
    // libzeth/tests/core/group_element_utils_test.cpp:42-52: components are written c1 first
    out_s << "[";
    field_element_write_json(el.c1, out_s);
    out_s << ",";
    field_element_write_json(el.c0, out_s);
    out_s << "]";
}

// ---- group elements (group_element_utils.tcc:93-113) ----
namespace internal
{
template<typename CoordT> bool coordinate_equals_zero(const CoordT &f) { return f.is_zero(); }
template<typename CoordT> bool coordinate_equals_one(const CoordT &f) { return f == CoordT::one(); }
} // namespace internal

template<typename CoordT, int Gid> void group_element_write_bytes(const group_element<CoordT, Gid> &point, std::ostream &out_s)
{
    if (point.is_zero()) {
        // libff's affine form of the identity is (0, 1)
        field_element_write_bytes(CoordT::zero(), out_s);
        field_element_write_bytes(CoordT::one(), out_s);
        return;
    }
    field_element_write_bytes(point.X, out_s);
    field_element_write_bytes(point.Y, out_s);
}

template<typename CoordT, int Gid> void group_element_read_bytes(group_element<CoordT, Gid> &point, std::istream &in_s)
{
    field_element_read_bytes(point.X, in_s);
    field_element_read_bytes(point.Y, in_s);
    if (internal::coordinate_equals_zero(point.X) && internal::coordinate_equals_one(point.Y)) {
        point = group_element<CoordT, Gid>::zero();
    }
}

template<typename CoordT, int Gid> void group_element_write_json(const group_element<CoordT, Gid> &point, std::ostream &out_s)
{
    out_s << "[";
    if (point.is_zero()) {
        field_element_write_json(CoordT::zero(), out_s);
        out_s << ",";
        field_element_write_json(CoordT::one(), out_s);
    } else {
        field_element_write_json(point.X, out_s);
        out_s << ",";
        field_element_write_json(point.Y, out_s);
    }
    out_s << "]";
}

template<typename GroupT> std::string group_element_to_json(const GroupT &point)
{
    std::stringstream ss;
    group_element_write_json(point, ss);
    return ss.str();
}

// ---- collections ----
namespace internal
{
// never trust a length prefix with a reservation larger than this many elements
constexpr size_t max_reserve = size_t(1) << 24;
} // namespace internal

template<typename GroupT> void group_elements_write_bytes(const std::vector<GroupT> &points, std::ostream &out_s)
{
    write_bytes(points.size(), out_s);
    for (const GroupT &p : points) {
        group_element_write_bytes(p, out_s);
    }
}

template<typename GroupT> void group_elements_read_bytes(std::vector<GroupT> &points, std::istream &in_s)
{
    const size_t n = read_bytes<size_t>(in_s);
    points.clear();
    points.reserve(n < internal::max_reserve ? n : internal::max_reserve);
    for (size_t i = 0; i < n; ++i) {
        points.emplace_back();
        group_element_read_bytes(points.back(), in_s);
    }
}

template<typename kcvectorT> void knowledge_commitment_vector_write_bytes(const kcvectorT &v, std::ostream &out_s)
{
    const size_t num_entries = v.indices.size();
    if (num_entries != v.values.size()) {
        throw std::invalid_argument("sparse vector: indices / values size mismatch");
    }
    write_bytes(v.domain_size_, out_s);
    write_bytes(num_entries, out_s);
    for (size_t i = 0; i < num_entries; ++i) {
        write_bytes(v.indices[i], out_s);
        group_element_write_bytes(v.values[i].g, out_s);
        group_element_write_bytes(v.values[i].h, out_s);
    }
}

template<typename kcvectorT> void knowledge_commitment_vector_read_bytes(kcvectorT &v, std::istream &in_s)
{
    v.domain_size_ = read_bytes<size_t>(in_s);
    const size_t num_entries = read_bytes<size_t>(in_s);
    v.indices.clear();
    v.values.clear();
    v.indices.reserve(num_entries < internal::max_reserve ? num_entries : internal::max_reserve);
    v.values.reserve(num_entries < internal::max_reserve ? num_entries : internal::max_reserve);
    for (size_t i = 0; i < num_entries; ++i) {
        v.indices.push_back(read_bytes<size_t>(in_s));
        v.values.emplace_back();
        group_element_read_bytes(v.values.back().g, in_s);
        group_element_read_bytes(v.values.back().h, in_s);
    }
}

/// sparse_vector / accumulation_vector of group elements (stream_utils.tcc:92-139)
template<typename GroupT> void sparse_vector_write_bytes(const sparse_vector<GroupT> &v, std::ostream &out_s)
{
    const size_t num_entries = v.indices.size();
    if (num_entries != v.values.size()) {
        throw std::invalid_argument("sparse vector: indices / values size mismatch");
    }
    write_bytes(v.domain_size_, out_s);
    write_bytes(num_entries, out_s);
    for (size_t i = 0; i < num_entries; ++i) {
        write_bytes(v.indices[i], out_s);
        group_element_write_bytes(v.values[i], out_s);
    }
}

template<typename GroupT> void sparse_vector_read_bytes(sparse_vector<GroupT> &v, std::istream &in_s)
{
    v.domain_size_ = read_bytes<size_t>(in_s);
    const size_t num_entries = read_bytes<size_t>(in_s);
    v.indices.clear();
    v.values.clear();
    v.indices.reserve(num_entries < internal::max_reserve ? num_entries : internal::max_reserve);
    v.values.reserve(num_entries < internal::max_reserve ? num_entries : internal::max_reserve);
    for (size_t i = 0; i < num_entries; ++i) {
        v.indices.push_back(read_bytes<size_t>(in_s));
        v.values.emplace_back();
        group_element_read_bytes(v.values.back(), in_s);
    }
}

template<typename GroupT> void accumulation_vector_write_bytes(const accumulation_vector<GroupT> &v, std::ostream &out_s)
{
    group_element_write_bytes(v.first, out_s);
    sparse_vector_write_bytes(v.rest, out_s);
}

template<typename GroupT> void accumulation_vector_read_bytes(accumulation_vector<GroupT> &v, std::istream &in_s)
{
    group_element_read_bytes(v.first, in_s);
    sparse_vector_read_bytes(v.rest, in_s);
}

// ---- r1cs (r1cs_serialization.tcc:199-275) ----
template<typename FieldT> void linear_combination_read_bytes(linear_combination<FieldT> &lc, std::istream &in_s)
{
    const uint32_t num_terms = read_bytes<uint32_t>(in_s);
    lc.terms.clear();
    lc.terms.reserve(num_terms < internal::max_reserve ? num_terms : internal::max_reserve);
    for (uint32_t i = 0; i < num_terms; ++i) {
        linear_term<FieldT> t;
        t.index = read_bytes<var_index_t>(in_s);
        field_element_read_bytes(t.coeff, in_s);
        lc.terms.push_back(t);
    }
}

template<typename FieldT> void linear_combination_write_bytes(const linear_combination<FieldT> &lc, std::ostream &out_s)
{
    if (lc.terms.size() > (size_t)std::numeric_limits<uint32_t>::max()) {
        throw std::length_error("linear combination with more than 2^32 - 1 terms");
    }
    write_bytes((uint32_t)lc.terms.size(), out_s);
    for (const linear_term<FieldT> &t : lc.terms) {
        write_bytes(t.index, out_s);
        field_element_write_bytes(t.coeff, out_s);
    }
}

template<typename FieldT> void r1cs_read_bytes(r1cs_constraint_system<FieldT> &r1cs, std::istream &in_s)
{
    r1cs.primary_input_size = read_bytes<size_t>(in_s);
    r1cs.auxiliary_input_size = read_bytes<size_t>(in_s);
    const size_t n = read_bytes<size_t>(in_s);
    r1cs.constraints.clear();
    r1cs.constraints.reserve(n < internal::max_reserve ? n : internal::max_reserve);
    for (size_t i = 0; i < n; ++i) {
        r1cs.constraints.emplace_back();
        linear_combination_read_bytes(r1cs.constraints.back().a, in_s);
        linear_combination_read_bytes(r1cs.constraints.back().b, in_s);
        linear_combination_read_bytes(r1cs.constraints.back().c, in_s);
    }
}

template<typename FieldT> void r1cs_write_bytes(const r1cs_constraint_system<FieldT> &r1cs, std::ostream &out_s)
{
    write_bytes(r1cs.primary_input_size, out_s);
    write_bytes(r1cs.auxiliary_input_size, out_s);
    write_bytes(r1cs.constraints.size(), out_s);
    for (const r1cs_constraint<FieldT> &c : r1cs.constraints) {
        linear_combination_write_bytes(c.a, out_s);
        linear_combination_write_bytes(c.b, out_s);
        linear_combination_write_bytes(c.c, out_s);
    }
}

// ---- assignments (r1cs_variable_assignment_serialization.tcc:15-77) ----
template<typename FieldT> void r1cs_variable_assignment_read_bytes(r1cs_variable_assignment<FieldT> &assignment, std::istream &in_s)
{
    const size_t n = read_bytes<size_t>(in_s);
    assignment.clear();
    assignment.reserve(n < internal::max_reserve ? n : internal::max_reserve);
    for (size_t i = 0; i < n; ++i) {
        assignment.emplace_back();
        field_element_read_bytes(assignment.back(), in_s);
    }
}

template<typename FieldT>
void r1cs_variable_assignment_read_bytes(
    r1cs_primary_input<FieldT> &primary, r1cs_auxiliary_input<FieldT> &auxiliary, const size_t primary_input_size, std::istream &in_s)
{
    const size_t n = read_bytes<size_t>(in_s);
    if (n < primary_input_size) {
        throw std::length_error("assignment length smaller than primary input size");
    }
    primary.clear();
    for (size_t i = 0; i < primary_input_size; ++i) {
        primary.emplace_back();
        field_element_read_bytes(primary.back(), in_s);
    }
    auxiliary.clear();
    auxiliary.reserve(n - primary_input_size < internal::max_reserve ? n - primary_input_size : internal::max_reserve);
    for (size_t i = primary_input_size; i < n; ++i) {
        auxiliary.emplace_back();
        field_element_read_bytes(auxiliary.back(), in_s);
    }
}

template<typename FieldT> void r1cs_variable_assignment_write_bytes(const r1cs_variable_assignment<FieldT> &assignment, std::ostream &out_s)
{
    write_bytes(assignment.size(), out_s);
    for (const FieldT &v : assignment) {
        field_element_write_bytes(v, out_s);
    }
}

template<typename FieldT>
void r1cs_variable_assignment_write_bytes(
    const r1cs_primary_input<FieldT> &primary, const r1cs_auxiliary_input<FieldT> &auxiliary, std::ostream &out_s)
{
    write_bytes(primary.size() + auxiliary.size(), out_s);
    for (const FieldT &v : primary) {
        field_element_write_bytes(v, out_s);
    }
    for (const FieldT &v : auxiliary) {
        field_element_write_bytes(v, out_s);
    }
}

} // namespace libzeth_gpu

#endif // LIBZETH_GPU_SERIALIZATION_HPP
