// Driver for the tests of the C++ host module (include/libzeth_gpu/): each subcommand exercises one piece of the
// adapter on files the Python tests prepare and check against the oracle.  Exit codes name the exception class so
// the error convention can be tested: 10 invalid_argument, 11 length_error, 12 ios_base::failure, 13 runtime_error.
//
//   codec CURVE pk_in pk_out assignment_in assignment_out [primary_inputs]     (no GPU)
//   testdata CURVE out_dir            the fixtures of libzeth/tests/snarks/groth16/groth16_snark_test.cpp:39-171 (no GPU compute)
//   multi_exp CURVE g1|g2 points scalars out_point
//   domain CURVE fft|ifft|cosetfft|icosetfft scalars_in scalars_out
//   lagrange CURVE g1|g2 powers out_points      evaluator_from_lagrange for every single-index factor map (evaluator_from_lagrange_test.cpp:44-65)
//   prove CURVE pk assignment primary_inputs r_hex s_hex proof_fixed proof_random_1 proof_random_2
//   multi_exp_sharded CURVE g1|g2 points scalars out_point n_shards     sharded_bases over n_shards contexts on device 0
//   prove_pool CURVE pk assignment primary_inputs r_hex s_hex proof_a proof_b     device_proving_key_pool over {0, 0}
//   prove_concurrent CURVE pk assignment primary_inputs r_hex s_hex n_threads proofs_per_thread out_prefix
//          n_threads callers share ONE device_proving_key_pool (device 0, 4 proofs in flight); proof k of the run uses
//          (r + k, s + 2k + 1) and is written to out_prefix.k; then the same threads call the stateless
//          generate_proof(pk, primary, auxiliary) concurrently (out_prefix.sK, fresh randomness).  Prints the peak number
//          of proofs that were in flight.
//   bench_pool CURVE r1cs assignment primary_inputs n_threads proofs_per_thread in_flight proof_out
//          key from fixed secrets (generator_from_secrets), then proofs/s through device_proving_key_pool; one JSON line
//   keypair CURVE keypair_in pk_out vk_out vk_json keypair_out     keypair_read_bytes, then every writer (no GPU)
//   swap_ab CURVE r1cs_in r1cs_out       r1cs_constraint_system::swap_AB_if_beneficial (no GPU); prints swapped / kept
//   setup CURVE r1cs t_hex alpha_hex beta_hex delta_hex pk_out vk_out proof_out assignment primary_inputs
//          r1cs_gg_ppzksnark_generator_from_secrets (gpu_groth16_generator.hpp) written as pk_out / vk_out and as
//          pk_out.keypair (keypair_write_bytes); then generate_setup with fresh secrets: vk_out.fresh and a proof made
//          with that key in proof_out
//   prove_pb CURVE pk assignment primary_inputs pk2 proof_1 proof_2 proof_3
//          the FIRST library call is generate_proof(pk, pb) (groth16_snark.hpp:32-35); then the host key is overwritten in
//          place with pk2 (same address, same sizes) and proved again: the stale device copy must not be used; then the
//          device keys are released and the key proved once more; the process must exit cleanly
#include <atomic>
#include <chrono>
#include <iostream>
#include <thread>

#include "../../include/libzeth_gpu/gpu_groth16_generator.hpp"
#include "../../include/libzeth_gpu/gpu_groth16_snark.hpp"

using namespace libzeth_gpu;

namespace
{

template<typename ppT> Fr<ppT> neg_fr(uint64_t k)
{
    // r - k for small k
    Fr<ppT> x;
    uint64_t borrow = k;
    for (size_t i = 0; i < 4; i++) {
        const uint64_t limb = ppT::r[i];
        x.data[i] = limb - borrow;
        borrow = limb < borrow ? 1 : 0;
    }
    return x;
}

template<typename ppT, typename GroupT> GroupT gen_multiple(const GroupT &gen, const Fr<ppT> &k)
{
    GroupT out;
    check(zg_host_group_op(ppT::curve_id, GroupT::group_id, 1, &gen, k.data, ZG_FORM_PLAIN, &out));
    return out;
}

int exception_code(const std::exception &e)
{
    if (dynamic_cast<const std::invalid_argument *>(&e)) {
        return 10;
    }
    if (dynamic_cast<const std::length_error *>(&e)) {
        return 11;
    }
    if (dynamic_cast<const std::ios_base::failure *>(&e)) {
        return 12;
    }
    return 13;
}

template<typename ppT> int cmd_codec(int argc, char **argv)
{
    using snarkT = groth16_snark<ppT>;
    typename snarkT::proving_key pk;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        snarkT::proving_key_read_bytes(pk, in_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(argv[4]);
        snarkT::proving_key_write_bytes(pk, out_s);
    }
    const size_t primary_inputs = argc > 7 ? std::stoul(argv[7]) : 1;
    r1cs_primary_input<Fr<ppT>> primary;
    r1cs_auxiliary_input<Fr<ppT>> auxiliary;
    {
        std::ifstream in_s = open_binary_input_file(argv[5]);
        r1cs_variable_assignment_read_bytes(primary, auxiliary, primary_inputs, in_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(argv[6]);
        r1cs_variable_assignment_write_bytes(primary, auxiliary, out_s);
    }
    std::cout << "codec ok: " << pk.A_query.size() << " A, " << pk.B_query.values.size() << " B, " << pk.H_query.size() << " H, "
              << pk.L_query.size() << " L, " << pk.constraint_system.num_constraints() << " constraints, " << primary.size() << "+"
              << auxiliary.size() << " assignment\n";
    return 0;
}

template<typename ppT, typename GroupT> std::vector<GroupT> read_points(const char *path)
{
    std::vector<GroupT> pts;
    std::ifstream in_s = open_binary_input_file(path);
    group_elements_read_bytes(pts, in_s);
    return pts;
}

template<typename ppT> std::vector<Fr<ppT>> read_scalars(const char *path)
{
    std::vector<Fr<ppT>> v;
    std::ifstream in_s = open_binary_input_file(path);
    r1cs_variable_assignment_read_bytes(v, in_s);
    return v;
}

template<typename ppT, typename GroupT> int cmd_multi_exp(char **argv)
{
    const std::vector<GroupT> gs = read_points<ppT, GroupT>(argv[4]);
    const std::vector<Fr<ppT>> fs = read_scalars<ppT>(argv[5]);
    const GroupT sum = multi_exp<ppT, GroupT>(gs, fs);
    // the same through resident bases and through the iterator overload must agree
    resident_bases<ppT, GroupT> rb(gs);
    if (rb.multi_exp(fs) != sum) {
        std::cerr << "resident_bases::multi_exp disagrees with multi_exp\n";
        return 1;
    }
    rb.precompute(12);  // the same through a fixed-base table
    if (rb.multi_exp(fs) != sum) {
        std::cerr << "resident_bases::multi_exp with a table disagrees with multi_exp\n";
        return 1;
    }
    if (!fs.empty() && multi_exp<ppT, GroupT>(gs.begin(), gs.begin() + fs.size(), fs.begin(), fs.end()) != sum) {
        std::cerr << "iterator multi_exp disagrees with multi_exp\n";
        return 1;
    }
    std::ofstream out_s = open_binary_output_file(argv[6]);
    group_element_write_bytes(sum, out_s);
    return 0;
}

template<typename ppT, typename GroupT> int cmd_multi_exp_sharded(char **argv)
{
    const std::vector<GroupT> gs = read_points<ppT, GroupT>(argv[4]);
    const std::vector<Fr<ppT>> fs = read_scalars<ppT>(argv[5]);
    const std::vector<int> devices(std::stoul(argv[7]), 0); // every shard on device 0, each with a context of its own
    sharded_bases<ppT, GroupT> sb(gs, devices);
    const GroupT sum = sb.multi_exp(fs);
    sb.precompute(12);
    if (sb.multi_exp(fs) != sum) {
        std::cerr << "sharded_bases::multi_exp with tables disagrees with the plain form\n";
        return 1;
    }
    std::ofstream out_s = open_binary_output_file(argv[6]);
    group_element_write_bytes(sum, out_s);
    return 0;
}

template<typename ppT> int cmd_domain(char **argv)
{
    std::vector<Fr<ppT>> a = read_scalars<ppT>(argv[4]);
    basic_radix2_domain<ppT> domain(a.size());
    const std::string op = argv[3];
    const Fr<ppT> g(ppT::multiplicative_generator);
    if (op == "fft") {
        domain.FFT(a);
    } else if (op == "ifft") {
        domain.iFFT(a);
    } else if (op == "cosetfft") {
        domain.cosetFFT(a, g);
    } else if (op == "icosetfft") {
        domain.icosetFFT(a, g);
    } else {
        throw std::invalid_argument("unknown transform");
    }
    std::ofstream out_s = open_binary_output_file(argv[5]);
    r1cs_variable_assignment_write_bytes(a, out_s);
    return 0;
}

template<typename ppT, typename GroupT> int cmd_lagrange(char **argv)
{
    const std::vector<GroupT> powers = read_points<ppT, GroupT>(argv[4]);
    basic_radix2_domain<ppT> domain(powers.size());
    evaluator_from_lagrange<ppT, GroupT> eval(powers, domain);
    std::vector<GroupT> out;
    for (size_t j = 0; j < domain.m; ++j) {
        std::map<size_t, Fr<ppT>> l_factors;
        l_factors[j] = Fr<ppT>::one();
        out.push_back(eval.evaluate_from_lagrange_factors(l_factors));
    }
    // the comparison the reference test makes: against the Lagrange evaluations from the group-element FFT
    std::vector<GroupT> lagrange = powers;
    compute_lagrange_from_powers<ppT, GroupT>(lagrange);
    for (size_t j = 0; j < domain.m; ++j) {
        if (out[j] != lagrange[j]) {
            std::cerr << "L_" << j << ": evaluator_from_lagrange and compute_lagrange_from_powers disagree\n";
            return 1;
        }
    }
    std::ofstream out_s = open_binary_output_file(argv[5]);
    group_elements_write_bytes(out, out_s);
    return 0;
}

template<size_t n> field_element<n> from_hex(const std::string &hex)
{
    field_element<n> v;
    size_t bit = 0;
    for (size_t i = hex.size(); i-- > 0 && bit < 64 * n;) {
        const char c = hex[i];
        const uint64_t d = (c >= '0' && c <= '9') ? c - '0' : (c >= 'a' && c <= 'f') ? c - 'a' + 10 : 0;
        v.data[bit / 64] |= d << (bit % 64);
        bit += 4;
    }
    return v;
}

template<typename ppT> int cmd_prove(char **argv)
{
    using snarkT = groth16_snark<ppT>;
    typename snarkT::proving_key pk;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        snarkT::proving_key_read_bytes(pk, in_s);
    }
    r1cs_primary_input<Fr<ppT>> primary;
    r1cs_auxiliary_input<Fr<ppT>> auxiliary;
    {
        std::ifstream in_s = open_binary_input_file(argv[4]);
        r1cs_variable_assignment_read_bytes(primary, auxiliary, std::stoul(argv[5]), in_s);
    }
    {
        device_proving_key<ppT> dpk(pk);
        const typename snarkT::proof proof = dpk.prove(primary, auxiliary, from_hex<4>(argv[6]), from_hex<4>(argv[7]));
        std::ofstream out_s = open_binary_output_file(argv[8]);
        snarkT::proof_write_bytes(proof, out_s);
    }
    // the reference's stateless entry point, twice: the second call reuses the resident key; (r, s) are fresh
    for (int k = 0; k < 2; k++) {
        const typename snarkT::proof proof = snarkT::generate_proof(pk, primary, auxiliary);
        std::ofstream out_s = open_binary_output_file(argv[9 + k]);
        snarkT::proof_write_bytes(proof, out_s);
    }
    // wrong sizes are invalid_argument, as a mismatched assignment is in the reference's prover
    bool threw = false;
    try {
        r1cs_auxiliary_input<Fr<ppT>> short_aux(auxiliary.begin(), auxiliary.end() - 1);
        snarkT::generate_proof(pk, primary, short_aux);
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    if (!threw) {
        std::cerr << "a short assignment did not raise invalid_argument\n";
        return 1;
    }
    return 0;
}

template<typename ppT> int cmd_prove_pool(char **argv)
{
    using snarkT = groth16_snark<ppT>;
    typename snarkT::proving_key pk;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        snarkT::proving_key_read_bytes(pk, in_s);
    }
    r1cs_primary_input<Fr<ppT>> primary;
    r1cs_auxiliary_input<Fr<ppT>> auxiliary;
    {
        std::ifstream in_s = open_binary_input_file(argv[4]);
        r1cs_variable_assignment_read_bytes(primary, auxiliary, std::stoul(argv[5]), in_s);
    }
    device_proving_key_pool<ppT> pool(pk, {0, 0}, 1); // two resident copies, one prover each; on a node: one per device
    for (int k = 0; k < 2; k++) {
        const typename snarkT::proof proof = pool.prove(primary, auxiliary, from_hex<4>(argv[6]), from_hex<4>(argv[7]));
        std::ofstream out_s = open_binary_output_file(argv[8 + k]);
        snarkT::proof_write_bytes(proof, out_s);
    }
    return 0;
}

template<typename ppT> Fr<ppT> add_small(Fr<ppT> x, uint64_t k)
{
    for (size_t i = 0; i < 4 && k; i++) {
        const uint64_t before = x.data[i];
        x.data[i] += k;
        k = x.data[i] < before ? 1 : 0;
    }
    return x; // callers keep x + k below r
}

template<typename ppT> int cmd_prove_concurrent(char **argv)
{
    using snarkT = groth16_snark<ppT>;
    typename snarkT::proving_key pk;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        snarkT::proving_key_read_bytes(pk, in_s);
    }
    r1cs_primary_input<Fr<ppT>> primary;
    r1cs_auxiliary_input<Fr<ppT>> auxiliary;
    {
        std::ifstream in_s = open_binary_input_file(argv[4]);
        r1cs_variable_assignment_read_bytes(primary, auxiliary, std::stoul(argv[5]), in_s);
    }
    const Fr<ppT> r0 = from_hex<4>(argv[6]), s0 = from_hex<4>(argv[7]);
    const size_t n_threads = std::stoul(argv[8]), per_thread = std::stoul(argv[9]);
    const std::string prefix = argv[10];
    device_proving_key_pool<ppT> pool(pk, {0}, 4);
    std::vector<typename snarkT::proof> proofs(n_threads * per_thread), stateless(n_threads);
    std::vector<std::string> errors(n_threads);
    std::atomic<size_t> ready(0);
    auto worker = [&](size_t t) {
        try {
            ready++;
            while (ready.load() < n_threads) { // all callers start together
                std::this_thread::yield();
            }
            for (size_t j = 0; j < per_thread; j++) {
                const size_t k = t * per_thread + j;
                proofs[k] = pool.prove(primary, auxiliary, add_small<ppT>(r0, k), add_small<ppT>(s0, 2 * k + 1));
            }
            stateless[t] = snarkT::generate_proof(pk, primary, auxiliary);
        } catch (const std::exception &e) {
            errors[t] = e.what();
        }
    };
    std::vector<std::thread> threads;
    for (size_t t = 0; t < n_threads; t++) {
        threads.emplace_back(worker, t);
    }
    for (auto &th : threads) {
        th.join();
    }
    for (size_t t = 0; t < n_threads; t++) {
        if (!errors[t].empty()) {
            std::cerr << "thread " << t << ": " << errors[t] << "\n";
            return 1;
        }
    }
    for (size_t k = 0; k < proofs.size(); k++) {
        std::ofstream out_s = open_binary_output_file(prefix + "." + std::to_string(k));
        snarkT::proof_write_bytes(proofs[k], out_s);
    }
    for (size_t t = 0; t < n_threads; t++) {
        std::ofstream out_s = open_binary_output_file(prefix + ".s" + std::to_string(t));
        snarkT::proof_write_bytes(stateless[t], out_s);
    }
    std::cout << "peak_in_flight " << pool.peak_in_flight() << " proofs " << pool.proofs_completed() << "\n";
    {
        // warm_up(): every prover of a fresh pool exists and has proved once, and proves the same proof afterwards
        device_proving_key_pool<ppT> warm(pk, {0}, 3);
        warm.warm_up(primary, auxiliary);
        const bool same = warm.prove(primary, auxiliary, add_small<ppT>(r0, 0), add_small<ppT>(s0, 1)) == proofs[0];
        std::cout << "warm_up provers " << warm.peak_in_flight() << " proofs " << warm.proofs_completed() << " same " << (same ? 1 : 0) << "\n";
    }
    return 0;
}

template<typename ppT> int cmd_bench_pool(char **argv)
{
    using snarkT = groth16_snark<ppT>;
    using clock = std::chrono::steady_clock;
    r1cs_constraint_system<Fr<ppT>> cs;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        r1cs_read_bytes(cs, in_s);
    }
    r1cs_primary_input<Fr<ppT>> primary;
    r1cs_auxiliary_input<Fr<ppT>> auxiliary;
    {
        std::ifstream in_s = open_binary_input_file(argv[4]);
        r1cs_variable_assignment_read_bytes(primary, auxiliary, std::stoul(argv[5]), in_s);
    }
    const size_t n_threads = std::stoul(argv[6]), per_thread = std::stoul(argv[7]), in_flight = std::stoul(argv[8]);
    auto t0 = clock::now();
    // the secrets of tools/bench_prove.py, so that the proof can be compared with the oracle's on the same key
    const r1cs_gg_ppzksnark_keypair<ppT> kp = r1cs_gg_ppzksnark_generator_from_secrets<ppT>(
        cs, from_hex<4>("5a4554485f746175"), from_hex<4>("616c706861"), from_hex<4>("62657461"), from_hex<4>("64656c7461"));
    const double keygen_s = std::chrono::duration<double>(clock::now() - t0).count();
    t0 = clock::now();
    device_proving_key_pool<ppT> pool(kp.pk, {0}, in_flight);
    const Fr<ppT> r = from_hex<4>("1234567890abcdef1234567890abcdef0fedcba9876543210fedcba987654321"),
                  s = from_hex<4>("0fedcba0987654321fedcba09876543211234567890abcdef1234567890abcde");
    const typename snarkT::proof first = pool.prove(primary, auxiliary, r, s);
    const double upload_s = std::chrono::duration<double>(clock::now() - t0).count();
    {
        std::ofstream out_s = open_binary_output_file(argv[9]);
        snarkT::proof_write_bytes(first, out_s);
    }
    // serial latency through the pool (one caller)
    std::vector<double> lat;
    for (int i = 0; i < 20; i++) {
        const auto a = clock::now();
        if (!(pool.prove(primary, auxiliary, r, s) == first)) {
            std::cerr << "serial proof differs\n";
            return 1;
        }
        lat.push_back(std::chrono::duration<double, std::milli>(clock::now() - a).count());
    }
    std::sort(lat.begin(), lat.end());
    t0 = clock::now();
    pool.warm_up(primary, auxiliary); // every prover exists and has proved once before the clock starts
    const double warm_up_s = std::chrono::duration<double>(clock::now() - t0).count();
    std::atomic<size_t> ready(0), bad(0);
    std::atomic<bool> go(false);
    auto worker = [&]() {
        ready++;
        while (!go.load()) {
            std::this_thread::yield();
        }
        for (size_t j = 0; j < per_thread; j++) {
            if (!(pool.prove(primary, auxiliary, r, s) == first)) {
                bad++;
            }
        }
    };
    std::vector<std::thread> threads;
    for (size_t t = 0; t < n_threads; t++) {
        threads.emplace_back(worker);
    }
    while (ready.load() < n_threads) {
        std::this_thread::yield();
    }
    const auto a = clock::now();
    go = true;
    for (auto &th : threads) {
        th.join();
    }
    const double wall = std::chrono::duration<double>(clock::now() - a).count();
    if (bad.load()) {
        std::cerr << bad.load() << " proofs differ from the serial proof\n";
        return 1;
    }
    std::cout << "{\"metric\": \"groth16_prove_through_device_proving_key_pool\", \"curve\": \"" << ppT::name << "\", \"constraints\": "
              << cs.num_constraints() << ", \"variables\": " << cs.num_variables() << ", \"callers\": " << n_threads
              << ", \"in_flight_per_device\": " << in_flight << ", \"peak_in_flight\": " << pool.peak_in_flight() << ", \"proofs\": "
              << n_threads * per_thread << ", \"wall_ms\": " << wall * 1e3 << ", \"proofs_per_s_per_gpu\": " << n_threads * per_thread / wall
              << ", \"ms_per_proof_median_one_caller\": " << lat[lat.size() / 2] << ", \"ms_per_proof_min_one_caller\": " << lat[0]
              << ", \"keygen_s\": " << keygen_s << ", \"upload_and_first_proof_s\": " << upload_s << ", \"warm_up_s\": " << warm_up_s << ", \"all_equal_to_serial_proof\": true}\n";
    return 0;
}

template<typename ppT> int cmd_keypair(char **argv)
{
    using snarkT = groth16_snark<ppT>;
    typename snarkT::keypair kp;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        snarkT::keypair_read_bytes(kp, in_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(argv[4]);
        snarkT::proving_key_write_bytes(kp.pk, out_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(argv[5]);
        snarkT::verification_key_write_bytes(kp.vk, out_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(argv[6]);
        snarkT::verification_key_write_json(kp.vk, out_s);
    }
    {
        // the verification key alone, read back through its own reader
        std::stringstream ss;
        snarkT::verification_key_write_bytes(kp.vk, ss);
        typename snarkT::verification_key vk2;
        snarkT::verification_key_read_bytes(vk2, ss);
        kp.vk = vk2;
    }
    {
        std::ofstream out_s = open_binary_output_file(argv[7]);
        snarkT::keypair_write_bytes(kp, out_s);
    }
    std::cout << "keypair ok: " << kp.pk.A_query.size() << " A, " << kp.vk.ABC_g1.size() << " ABC\n";
    return 0;
}

template<typename ppT> int cmd_swap_ab(char **argv)
{
    r1cs_constraint_system<Fr<ppT>> cs;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        r1cs_read_bytes(cs, in_s);
    }
    const bool swapped = cs.swap_AB_if_beneficial();
    {
        std::ofstream out_s = open_binary_output_file(argv[4]);
        r1cs_write_bytes(cs, out_s);
    }
    std::cout << (swapped ? "swapped" : "kept") << "\n";
    return 0;
}

template<typename ppT> int cmd_prove_pb(char **argv)
{
    using snarkT = groth16_snark<ppT>;
    typename snarkT::proving_key pk;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        snarkT::proving_key_read_bytes(pk, in_s);
    }
    protoboard<Fr<ppT>> pb;
    {
        r1cs_primary_input<Fr<ppT>> primary;
        r1cs_auxiliary_input<Fr<ppT>> auxiliary;
        std::ifstream in_s = open_binary_input_file(argv[4]);
        r1cs_variable_assignment_read_bytes(primary, auxiliary, std::stoul(argv[5]), in_s);
        pb.values = primary;
        pb.values.insert(pb.values.end(), auxiliary.begin(), auxiliary.end());
        pb.set_input_sizes(primary.size());
    }
    auto write = [](const typename snarkT::proof &proof, const char *path) {
        std::ofstream out_s = open_binary_output_file(path);
        snarkT::proof_write_bytes(proof, out_s);
    };
    write(snarkT::generate_proof(pk, pb), argv[7]); // nothing of the library has run before this call
    {
        typename snarkT::proving_key pk2;
        std::ifstream in_s = open_binary_input_file(argv[6]);
        snarkT::proving_key_read_bytes(pk2, in_s);
        pk.alpha_g1 = pk2.alpha_g1; // in place: the address the cache knows now holds another key
        pk.beta_g1 = pk2.beta_g1;
        pk.beta_g2 = pk2.beta_g2;
        pk.delta_g1 = pk2.delta_g1;
        pk.delta_g2 = pk2.delta_g2;
        std::copy(pk2.A_query.begin(), pk2.A_query.end(), pk.A_query.begin());
        std::copy(pk2.B_query.values.begin(), pk2.B_query.values.end(), pk.B_query.values.begin());
        std::copy(pk2.H_query.begin(), pk2.H_query.end(), pk.H_query.begin());
        std::copy(pk2.L_query.begin(), pk2.L_query.end(), pk.L_query.begin());
    }
    write(snarkT::generate_proof(pk, pb), argv[8]);
    release_device_keys<ppT>();
    write(snarkT::generate_proof(pk, pb), argv[9]);
    return 0; // the remaining device key is released by the exit hook
}

// groth16_snark_test.cpp:39-171 (generate_test_data): the hand-made constraint system, proving key and assignment
// setup <curve> <r1cs> <t> <alpha> <beta> <delta> <pk out> <vk out> <proof out> <assignment> <num inputs>:
// the key from explicit secrets (mpc_test.cpp:344-365 compares the ceremony's key with this function's), written in the
// reference's on-disk formats, and a proof made with that key from a randomly drawn setup as well
template<typename ppT> int cmd_setup(char **argv)
{
    using snarkT = groth16_snark<ppT>;
    r1cs_constraint_system<Fr<ppT>> cs;
    {
        std::ifstream in_s = open_binary_input_file(argv[3]);
        r1cs_read_bytes(cs, in_s);
    }
    const r1cs_gg_ppzksnark_keypair<ppT> kp =
        r1cs_gg_ppzksnark_generator_from_secrets<ppT>(cs, from_hex<4>(argv[4]), from_hex<4>(argv[5]), from_hex<4>(argv[6]), from_hex<4>(argv[7]));
    {
        std::ofstream out_s = open_binary_output_file(argv[8]);
        snarkT::proving_key_write_bytes(kp.pk, out_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(argv[9]);
        snarkT::verification_key_write_bytes(kp.vk, out_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(std::string(argv[8]) + ".keypair");
        snarkT::keypair_write_bytes(kp, out_s);
    }
    // generate_setup: fresh secrets; the key must prove (checked by the caller with the verification key written here)
    const r1cs_gg_ppzksnark_keypair<ppT> fresh = generate_setup<ppT>(cs);
    r1cs_primary_input<Fr<ppT>> primary;
    r1cs_auxiliary_input<Fr<ppT>> auxiliary;
    {
        std::ifstream in_s = open_binary_input_file(argv[11]);
        r1cs_variable_assignment_read_bytes(primary, auxiliary, std::stoul(argv[12]), in_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(argv[10]);
        snarkT::proof_write_bytes(snarkT::generate_proof(fresh.pk, primary, auxiliary), out_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(std::string(argv[9]) + ".fresh");
        snarkT::verification_key_write_bytes(fresh.vk, out_s);
    }
    return 0;
}

template<typename ppT> int cmd_testdata(char **argv)
{
    using Field = Fr<ppT>;
    using snarkT = groth16_snark<ppT>;
    const std::string dir = argv[3];
    const std::string curve_name = ppT::name;

    r1cs_constraint_system<Field> cs;
    linear_combination<Field> cs1_A, cs1_B, cs1_C, cs2_A, cs2_B, cs2_C;
    cs1_A.add_term(1, 2);
    cs1_A.add_term(3, 3);
    cs1_B.add_term(2, 4);
    cs1_B.add_term(4, 5);
    cs1_C.add_term(5, 6);
    cs1_C.add_term(1, 7);
    cs.add_constraint({cs1_A, cs1_B, cs1_C});
    cs2_A.add_term(2, 7);
    cs2_A.add_term(5, 6);
    cs2_B.add_term(1, 5);
    cs2_B.add_term(3, 4);
    cs2_C.add_term(4, 3);
    cs2_C.add_term(2, 2);
    cs.add_constraint({cs2_A, cs2_B, cs2_C});
    cs.primary_input_size = 2;
    cs.auxiliary_input_size = 3;
    {
        std::ofstream out_s = open_binary_output_file(dir + "/groth16_r1cs_" + curve_name);
        r1cs_write_bytes(cs, out_s);
    }

    // generators arrive as files written by the test (the library exports no generator constant through the ABI)
    const G1<ppT> g1 = read_points<ppT, G1<ppT>>((dir + "/g1_one").c_str()).at(0);
    const G2<ppT> g2 = read_points<ppT, G2<ppT>>((dir + "/g2_one").c_str()).at(0);
    auto P1 = [&](long k) { return gen_multiple<ppT, G1<ppT>>(g1, k >= 0 ? Field((uint64_t)k) : neg_fr<ppT>((uint64_t)(-k))); };
    auto P2 = [&](long k) { return gen_multiple<ppT, G2<ppT>>(g2, k >= 0 ? Field((uint64_t)k) : neg_fr<ppT>((uint64_t)(-k))); };

    typename snarkT::proving_key pk;
    pk.alpha_g1 = P1(1);
    pk.beta_g1 = P1(-1);
    pk.beta_g2 = P2(-1);
    pk.delta_g1 = P1(-2);
    pk.delta_g2 = P2(-2);
    pk.A_query = {P1(7), P1(-3), P1(8), P1(-4), P1(9), P1(-5)};
    const long b_mult[6] = {9, -9, 10, -11, 12, -13};
    // libsnark's knowledge_commitment_vector(std::vector) constructor indexes the entries 0..n-1 and sets
    // domain_size_ = n
    for (size_t i = 0; i < 6; i++) {
        pk.B_query.indices.push_back(i);
        pk.B_query.values.push_back({P2(b_mult[i]), P1(b_mult[i])});
    }
    pk.B_query.domain_size_ = 6;
    pk.H_query = {P1(11), P1(-11), P1(12)};
    pk.L_query = {P1(13), P1(-13), P1(14)};
    pk.constraint_system = cs;
    {
        std::ofstream out_s = open_binary_output_file(dir + "/groth16_proving_key_" + curve_name);
        snarkT::proving_key_write_bytes(pk, out_s);
    }

    const r1cs_variable_assignment<Field> assignment = {
        Field(15), neg_fr<ppT>(15), Field(16), neg_fr<ppT>(16), Field(17), neg_fr<ppT>(17)};
    {
        std::ofstream out_s = open_binary_output_file(dir + "/groth16_assignment_" + curve_name);
        r1cs_variable_assignment_write_bytes(assignment, out_s);
    }

    // proof round trip in the shape of common_snark_tests.tcc:102-122
    typename snarkT::proof proof{P1(21), P2(-21), P1(22)};
    std::stringstream ss;
    snarkT::proof_write_bytes(proof, ss);
    typename snarkT::proof back;
    snarkT::proof_read_bytes(back, ss);
    if (!(back == proof)) {
        std::cerr << "proof bytes round trip failed\n";
        return 1;
    }
    {
        std::ofstream out_s = open_binary_output_file(dir + "/groth16_proof_" + curve_name);
        snarkT::proof_write_bytes(proof, out_s);
    }
    {
        std::ofstream out_s = open_binary_output_file(dir + "/groth16_proof_" + curve_name + ".json");
        snarkT::proof_write_json(proof, out_s);
    }
    return 0;
}

template<typename ppT> int run(int argc, char **argv)
{
    const std::string cmd = argv[1];
    if (cmd == "codec" && argc >= 7) {
        return cmd_codec<ppT>(argc, argv);
    }
    if (cmd == "testdata" && argc >= 4) {
        return cmd_testdata<ppT>(argv);
    }
    if (cmd == "multi_exp" && argc >= 7) {
        return std::string(argv[3]) == "g1" ? cmd_multi_exp<ppT, G1<ppT>>(argv) : cmd_multi_exp<ppT, G2<ppT>>(argv);
    }
    if (cmd == "multi_exp_sharded" && argc >= 8) {
        return std::string(argv[3]) == "g1" ? cmd_multi_exp_sharded<ppT, G1<ppT>>(argv) : cmd_multi_exp_sharded<ppT, G2<ppT>>(argv);
    }
    if (cmd == "prove_pool" && argc >= 10) {
        return cmd_prove_pool<ppT>(argv);
    }
    if (cmd == "domain" && argc >= 6) {
        return cmd_domain<ppT>(argv);
    }
    if (cmd == "lagrange" && argc >= 6) {
        return std::string(argv[3]) == "g1" ? cmd_lagrange<ppT, G1<ppT>>(argv) : cmd_lagrange<ppT, G2<ppT>>(argv);
    }
    if (cmd == "prove" && argc >= 11) {
        return cmd_prove<ppT>(argv);
    }
    if (cmd == "setup" && argc >= 13) {
        return cmd_setup<ppT>(argv);
    }
    if (cmd == "prove_pb" && argc >= 10) {
        return cmd_prove_pb<ppT>(argv);
    }
    if (cmd == "prove_concurrent" && argc >= 11) {
        return cmd_prove_concurrent<ppT>(argv);
    }
    if (cmd == "bench_pool" && argc >= 10) {
        return cmd_bench_pool<ppT>(argv);
    }
    if (cmd == "keypair" && argc >= 8) {
        return cmd_keypair<ppT>(argv);
    }
    if (cmd == "swap_ab" && argc >= 5) {
        return cmd_swap_ab<ppT>(argv);
    }
    std::cerr << "usage: see the head of host_adapter_test.cpp\n";
    return 2;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::cerr << "usage: see the head of host_adapter_test.cpp\n";
        return 2;
    }
    try {
        const std::string curve = argv[2];
        if (curve == "alt-bn128") {
            return run<alt_bn128_pp>(argc, argv);
        }
        if (curve == "bls12-377") {
            return run<bls12_377_pp>(argc, argv);
        }
        std::cerr << "unrecognized curve\n";
        return 2;
    } catch (const std::exception &e) {
        std::cerr << "exception: " << e.what() << "\n";
        return exception_code(e);
    }
}
