"""krisp_amd -- MI355X-native drop-in for krisp_fasta's k-mer generation / sort /
multi-genome intersection path (reference: grunwaldlab/krisp @ 2024_10_08).

Python host code (this package) keeps the `kstream` generator surface and the
`krisp_fasta` command line; the hot path runs in hand-written HIP kernels
(csrc/krisp_hip.hip) behind the C ABI of include/krisp_hip.h, bound with ctypes.
"""
__version__ = "0.1.0"
