from gem_amd.embedding.lap import LaplacianEigenmaps  # noqa: F401
