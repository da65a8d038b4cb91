// alt_bn128 G1 instantiation of the MSM pipeline.
#include "group_ops.cuh"
using namespace zg;
using Gen = G1Gen<Bn254Fq, Bn254G1Params>;
ZG_DEFINE_GROUP_OPS(ZG_OPS_BN254_G1, Bn254Fq, Bn254Fr, Gen)
