from gem_amd.evaluation.reconstruction import evaluateStaticGraphReconstruction  # noqa: F401
