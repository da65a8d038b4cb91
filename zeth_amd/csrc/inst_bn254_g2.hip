// alt_bn128 G2 instantiation of the MSM pipeline.
#include "group_ops.cuh"
using namespace zg;
using Gen = G2Gen<Bn254Fq2, Bn254G2Params>;
ZG_DEFINE_GROUP_OPS(ZG_OPS_BN254_G2, Bn254Fq2, Bn254Fr, Gen)
