"""`gem` import-path alias of gem_amd, so GEM drivers written against `gem.embedding.*` import unchanged.
The three in-scope methods (HOPE, GraphFactorization, node2vec), Laplacian Eigenmaps and LLE (built on the HOPE solver), the
graph-reconstruction evaluator and the text wire formats are provided; SDNE is importable and raises NotImplementedError on construction-time use
(SURVEY 2 #11); plotting and the dynamic-graph utilities are out of scope and are not aliased."""
