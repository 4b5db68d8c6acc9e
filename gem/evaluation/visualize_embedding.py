"""Minimal 2-D scatter of an embedding for GEM-style drivers (plotting is out of scope of the backend, SURVEY section 2 #13;
this keeps `from gem.evaluation import visualize_embedding as viz` importable).  d > 2 is projected on its first two
principal components (upstream uses t-SNE)."""
import numpy as np


def plot_embedding2D(node_pos, node_colors=None, di_graph=None, labels=None):
    import matplotlib.pyplot as plt
    X = np.asarray(node_pos, dtype=float)
    if X.shape[1] > 2:
        Xc = X - X.mean(axis=0)
        _, _, vt = np.linalg.svd(Xc, full_matrices=False)
        X = Xc @ vt[:2].T
    if di_graph is not None:
        for i, j in di_graph.edges():
            plt.plot([X[i, 0], X[j, 0]], [X[i, 1], X[j, 1]], color='0.8', linewidth=0.5, zorder=1)
    plt.scatter(X[:, 0], X[:, 1], c=node_colors, s=25, zorder=2)
