from gem_amd.embedding.node2vec import node2vec  # noqa: F401
