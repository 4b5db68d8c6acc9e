from gem_amd.embedding.hope import HOPE  # noqa: F401
