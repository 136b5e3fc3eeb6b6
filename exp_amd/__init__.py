"""exp_amd: MI355X-native BFE N-body hot path (drop-in for EXP's sphereSL / cylinder force methods)."""
__version__ = "0.1.0"
