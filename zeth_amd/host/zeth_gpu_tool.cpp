// zeth-gpu-tool: the `prove` subcommand of zeth-tool (zeth_tool/prove_cmd.cpp:30-131) on the MI355X backend.
//
//   zeth-gpu-tool prove [-c|--curve alt-bn128|bls12-377] [-s|--snark groth16] [-p|--primary_inputs N]
//                       [-r|--profile] [pk_file] [assignment_file] [proof_file]
//
// Same files as the reference: proving key as written by groth16_snark::proving_key_write_bytes, assignment as
// written by r1cs_variable_assignment_write_bytes, proof written with proof_write_bytes.  Curve and snark options
// and their defaults follow zeth_tool/tool_common.hpp:44-67,103-116 (pghr13 is outside this backend).
// One extension for reproducible runs: --randomness R_HEX S_HEX fixes the prover's (r, s); by default they are drawn
// from the OS as libsnark does.
#include <chrono>
#include <iostream>

#include "../../include/libzeth_gpu/gpu_groth16_snark.hpp"

using namespace libzeth_gpu;

namespace
{

struct options {
    std::string curve = "alt-bn128";
    std::string snark = "groth16";
    std::string pk_file, assignment_file, proof_file;
    uint16_t num_primary_inputs = 1; // prove_cmd.cpp:25
    bool profile = false;
    bool fixed_randomness = false;
    std::string r_hex, s_hex;
};

void usage(const char *argv0)
{
    std::cout << "Usage:\n  " << argv0 << " prove [pk_file] [assignment_file] [proof_file]\n\n"
              << "Options:\n"
                 "  -c, --curve arg           Curve: alt-bn128 or bls12-377 (default: alt-bn128)\n"
                 "  -s, --snark arg           Snark: groth16 (default)\n"
                 "  -p, --primary_inputs arg  Number of primary inputs (default: 1)\n"
                 "  -r, --profile             Enable profiling output\n"
                 "      --randomness R S      Fix the prover randomness (hex, for reproducible proofs)\n";
}

template<size_t n> field_element<n> field_element_from_hex(std::string hex)
{
    if (hex.size() >= 2 && hex[0] == '0' && (hex[1] == 'x' || hex[1] == 'X')) {
        hex = hex.substr(2);
    }
    if (hex.empty() || hex.size() > 16 * n) {
        throw std::invalid_argument("invalid hex field element");
    }
    field_element<n> v;
    size_t bit = 0;
    for (size_t i = hex.size(); i-- > 0;) {
        const char c = hex[i];
        uint64_t d;
        if (c >= '0' && c <= '9') {
            d = c - '0';
        } else if (c >= 'a' && c <= 'f') {
            d = c - 'a' + 10;
        } else if (c >= 'A' && c <= 'F') {
            d = c - 'A' + 10;
        } else {
            throw std::invalid_argument("invalid hex field element");
        }
        v.data[bit / 64] |= d << (bit % 64);
        bit += 4;
    }
    return v;
}

double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

template<typename ppT> int prove(const options &o)
{
    using snarkT = groth16_snark<ppT>;
    ppT::init_public_params();

    auto t0 = std::chrono::steady_clock::now();
    typename snarkT::proving_key proving_key;
    {
        std::ifstream in_s = open_binary_input_file(o.pk_file);
        snarkT::proving_key_read_bytes(proving_key, in_s);
    }
    if (o.profile) {
        std::cout << "(profile) proving key read: " << ms_since(t0) << " ms\n";
    }

    r1cs_primary_input<Fr<ppT>> primary;
    r1cs_auxiliary_input<Fr<ppT>> auxiliary;
    {
        std::ifstream in_s = open_binary_input_file(o.assignment_file);
        r1cs_variable_assignment_read_bytes(primary, auxiliary, o.num_primary_inputs, in_s);
    }

    t0 = std::chrono::steady_clock::now();
    device_proving_key<ppT> device_pk(proving_key);
    if (o.profile) {
        std::cout << "(profile) proving key resident on device: " << ms_since(t0) << " ms\n";
    }

    t0 = std::chrono::steady_clock::now();
    typename snarkT::proof proof;
    if (o.fixed_randomness) {
        const Fr<ppT> r = field_element_from_hex<4>(o.r_hex), s = field_element_from_hex<4>(o.s_hex);
        if (!r.less_than(ppT::r) || !s.less_than(ppT::r)) {
            throw std::invalid_argument("--randomness: values must be below the group order");
        }
        proof = device_pk.prove(primary, auxiliary, r, s);
    } else {
        proof = device_pk.prove(primary, auxiliary, random_field_element<ppT>(), random_field_element<ppT>());
    }
    if (o.profile) {
        std::cout << "(profile) generate_proof: " << ms_since(t0) << " ms\n";
    }

    std::cout << "Writing proof to file: " << o.proof_file << "\n";
    {
        std::ofstream out_s = open_binary_output_file(o.proof_file);
        snarkT::proof_write_bytes(proof, out_s);
    }
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    options o;
    std::vector<std::string> positional;
    if (argc < 2 || std::string(argv[1]) == "-h" || std::string(argv[1]) == "--help" || std::string(argv[1]) == "help") {
        usage(argv[0]);
        return argc < 2 ? 1 : 0;
    }
    if (std::string(argv[1]) != "prove") {
        std::cerr << "error: unknown command: " << argv[1] << " (this tool implements `prove`)\n";
        return 1;
    }
    try {
        for (int i = 2; i < argc; i++) {
            const std::string a = argv[i];
            auto value = [&](const char *what) -> std::string {
                if (i + 1 >= argc) {
                    throw std::invalid_argument(std::string("missing value for ") + what);
                }
                return argv[++i];
            };
            if (a == "-c" || a == "--curve") {
                o.curve = value("--curve");
            } else if (a == "-s" || a == "--snark") {
                o.snark = value("--snark");
            } else if (a == "-p" || a == "--primary_inputs") {
                const unsigned long v = std::stoul(value("--primary_inputs"));
                if (v > 0xffff) {
                    throw std::invalid_argument("primary_inputs out of range");
                }
                o.num_primary_inputs = (uint16_t)v;
            } else if (a == "-r" || a == "--profile") {
                o.profile = true;
            } else if (a == "--randomness") {
                o.r_hex = value("--randomness");
                o.s_hex = value("--randomness");
                o.fixed_randomness = true;
            } else if (a == "-h" || a == "--help") {
                usage(argv[0]);
                return 0;
            } else if (!a.empty() && a[0] == '-') {
                throw std::invalid_argument("unrecognised option " + a);
            } else {
                positional.push_back(a);
            }
        }
        if (positional.size() < 1) {
            throw std::invalid_argument("pk_file not specified");
        }
        if (positional.size() < 2) {
            throw std::invalid_argument("assignment_file not specified");
        }
        if (positional.size() < 3) {
            throw std::invalid_argument("proof_file not specified");
        }
        if (positional.size() > 3) {
            throw std::invalid_argument("too many positional arguments");
        }
        o.pk_file = positional[0];
        o.assignment_file = positional[1];
        o.proof_file = positional[2];
        if (o.snark != "groth16") {
            throw std::invalid_argument(o.snark == "pghr13" ? "pghr13 is not provided by the GPU backend" : "unrecognized snark");
        }
        if (o.curve == "alt-bn128") {
            return prove<alt_bn128_pp>(o);
        } else if (o.curve == "bls12-377") {
            return prove<bls12_377_pp>(o);
        }
        throw std::invalid_argument("unrecognized curve");
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 1;
    }
}
