// BLS12-377 G2 instantiation of the MSM pipeline.
#include "group_ops.cuh"
using namespace zg;
using Gen = G2Gen<Bls377Fq2, Bls377G2Params>;
ZG_DEFINE_GROUP_OPS(ZG_OPS_BLS377_G2, Bls377Fq2, Bls377Fr, Gen)
