"""gem_amd -- MI355X-native backend for GEM's HOPE / GraphFactorization / node2vec
learn_embedding() hot path.  See DESIGN.md; the C ABI is include/gem_hip.h."""
__version__ = '0.1.0'
