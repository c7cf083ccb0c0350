"""MI355X-native Conformer-CTC training path: drop-in for NeMo's EncDecCTCModel / ConformerEncoder hot path.

Layout: csrc/ (HIP kernels + C ABI, built into lib/libmi355x_asr.so), ops.py (ctypes wrappers), modules/ + models/
(host-side mirror of the reference's NeuralModule interface), flat.py / packing.py / optim.py / parallel.py (flat
parameter storage, GEMM operand images, fused AdamW, RCCL gradient exchange)."""
__version__ = "0.1.0"
