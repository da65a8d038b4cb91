// Scalar-field (Fr) side of both curves: evaluation-domain transforms, witness map and prover assembly.
#include "prover.cuh"
using namespace zg;

static const FrOps OPS_BN254 = {&ntt_inplace<Bn254Fr>, &pk_create_impl<Bn254Fq, Bn254Fq2, Bn254Fr>, &ntt_static<Bn254Fr>,
                                &ntt_static_bytes};
static const FrOps OPS_BLS377 = {&ntt_inplace<Bls377Fr>, &pk_create_impl<Bls377Fq, Bls377Fq2, Bls377Fr>, &ntt_static<Bls377Fr>,
                                 &ntt_static_bytes};

const FrOps* zg_fr_ops(int curve) {
  if (curve == ZG_CURVE_ALT_BN128) return &OPS_BN254;
  if (curve == ZG_CURVE_BLS12_377) return &OPS_BLS377;
  return nullptr;
}
