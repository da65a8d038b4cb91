"""zeth_amd -- MI355X (gfx950) Groth16 proving backend for Zeth's prover hot path.

Host-side mirror of the reference's operator interface for the path (libzeth::multi_exp,
evaluation_domain FFTs, groth16_snark::generate_proof) over the C ABI in include/zeth_gpu.h.
There is no CPU fallback: importing works anywhere, calling needs the built HIP library and a GPU."""
from .lib import (  # noqa: F401
    ALT_BN128,
    BLS12_377,
    G1,
    G2,
    Bases,
    Context,
    DeviceBuffer,
    EvaluationDomain,
    Groth16Snark,
    ProvingKey,
    ZethGpuError,
    group_fft,
    group_scale,
    group_sparse_combine,
    host_group_op,
    load_library,
    msm_sharded,
    runtime_info,
)
