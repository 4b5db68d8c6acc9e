"""Text wire formats of GEM's native executables, kept for interoperability
(gem/utils/graph_util.py:129-169).  The HIP backend itself never touches text
files -- arrays cross the C ABI directly -- but the oracle drives the real
reference binaries (the real `gf` and `node2vec` executables) through these.

  saveGraphToEdgeListTxt      header `n`, `m`, then "%d %d %f"      (:129-134, input of `gf`)
  saveGraphToEdgeListTxtn2v   no header, "%d %d %f"                 (:137-140, input of `node2vec`)
  loadGraphFromEdgeListTxt    "i j [w]" lines -> nx graph           (:143-158)
  loadEmbedding               "n d" header, rows "id v1..vd", row placed at X[id]   (:161-169)
"""
import numpy as np


def _triples(graph):
    return graph.edges(data='weight', default=1)


def saveGraphToEdgeListTxt(graph, file_name):
    lines = ['%d %d %f\n' % (i, j, w) for i, j, w in _triples(graph)]
    with open(file_name, 'w') as fh:
        fh.write('%d\n%d\n' % (len(graph.nodes), len(lines)))
        fh.writelines(lines)


def saveGraphToEdgeListTxtn2v(graph, file_name):
    with open(file_name, 'w') as fh:
        fh.writelines('%d %d %f\n' % (i, j, w) for i, j, w in _triples(graph))


def loadGraphFromEdgeListTxt(file_name, directed=True):
    import networkx as nx
    G = nx.DiGraph() if directed else nx.Graph()
    with open(file_name) as fh:
        for line in fh:
            tok = line.split()
            if not tok:
                continue
            G.add_edge(int(tok[0]), int(tok[1]), weight=float(tok[2]) if len(tok) == 3 else 1.0)
    return G


def loadEmbedding(file_name):
    with open(file_name) as fh:
        n, d = (int(t) for t in fh.readline().split())
        X = np.zeros((n, d))
        for line in fh:
            tok = line.split()
            if tok:
                X[int(tok[0]), :] = [float(t) for t in tok[1:]]
    return X
