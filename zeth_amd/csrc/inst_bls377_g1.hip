// BLS12-377 G1 instantiation of the MSM pipeline.
#include "group_ops.cuh"
using namespace zg;
using Gen = G1Gen<Bls377Fq, Bls377G1Params>;
ZG_DEFINE_GROUP_OPS(ZG_OPS_BLS377_G1, Bls377Fq, Bls377Fr, Gen)
