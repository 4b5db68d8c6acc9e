from gem_amd.embedding.lle import LocallyLinearEmbedding  # noqa: F401
