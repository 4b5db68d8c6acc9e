"""`gem` import-path alias of gem_amd, so GEM drivers written against `gem.embedding.*` import unchanged.
Only the three in-scope methods, the graph-reconstruction evaluator and the text wire formats are provided
(SURVEY section 8); LE / LLE / SDNE / plotting are out of scope and are not aliased."""
