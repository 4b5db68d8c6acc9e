from gem_amd.embedding.static_graph_embedding import StaticGraphEmbedding  # noqa: F401
