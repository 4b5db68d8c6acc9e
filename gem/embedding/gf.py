from gem_amd.embedding.gf import GraphFactorization  # noqa: F401
