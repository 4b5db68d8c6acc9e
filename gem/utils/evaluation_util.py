from gem_amd.utils.evaluation_util import split_di_graph_to_train_test  # noqa: F401
