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


# ----------------------------------------------------------------------------------------------------------------------
# Array fast paths.  At 10M+ edges the per-line Python of the functions above (and the networkx graph
# they build) costs minutes on either side of a sub-second kernel; these read/write the SAME wire formats with the
# pandas C parser / vectorised formatting and hand back an EdgeListGraph (accepted by every learn_embedding()), plus a
# binary embedding container for tables that a text file would blow up 3x.

def loadEdgeListArrays(file_name, n=None, header=False):
    """'i j [w]' lines (the `node2vec` input; with header=True the `gf` input: first two lines n and m) -> EdgeListGraph.
    n defaults to the header's n, else max id + 1."""
    import pandas as pd
    from gem_amd.graph import EdgeListGraph
    skip = 0
    if header:
        with open(file_name) as fh:
            n_hdr, m_hdr = int(fh.readline()), int(fh.readline())
        skip, n = 2, (n_hdr if n is None else n)
    try:
        df = pd.read_csv(file_name, sep=r'\s+', header=None, skiprows=skip, engine='c')
    except pd.errors.EmptyDataError:
        df = pd.DataFrame({0: [], 1: []})
    src = df[0].to_numpy(dtype=np.int64)
    dst = df[1].to_numpy(dtype=np.int64)
    w = df[2].to_numpy(dtype=np.float32) if df.shape[1] >= 3 else None
    if header and len(src) != m_hdr:
        raise ValueError('%s: header announces %d edges, file holds %d' % (file_name, m_hdr, len(src)))
    if n is None:
        n = int(max(src.max(), dst.max())) + 1 if len(src) else 0
    if len(src) and (min(src.min(), dst.min()) < 0 or max(src.max(), dst.max()) >= n):
        raise ValueError('%s: node id outside [0, %d)' % (file_name, n))
    return EdgeListGraph(n, src, dst, w)


def saveEdgeListArrays(graph, file_name, header=False):
    """Write "%d %d %f" lines (byte-identical to saveGraphToEdgeListTxt[n2v]) from an EdgeListGraph / nx graph, vectorised."""
    import pandas as pd
    from gem_amd.graph import edge_arrays
    n, src, dst, w, _ = edge_arrays(graph)
    w = np.ones(len(src), dtype=np.float64) if w is None else np.asarray(w, dtype=np.float64)
    with open(file_name, 'w') as fh:
        if header:
            fh.write('%d\n%d\n' % (n, len(src)))
        pd.DataFrame({'i': src, 'j': dst, 'w': w}).to_csv(fh, sep=' ', header=False, index=False, float_format='%f', lineterminator='\n')


def saveEmbedding(X, file_name):
    """The text format both native executables emit and loadEmbedding parses: 'n d', then 'id v1 .. vd' per row."""
    X = np.asarray(X)
    with open(file_name, 'w') as fh:
        fh.write('%d %d\n' % X.shape)
        np.savetxt(fh, np.column_stack([np.arange(X.shape[0]), X]), fmt=['%d'] + ['%.8g'] * X.shape[1])


def loadEmbeddingFast(file_name):
    """loadEmbedding (rows placed by id, absent ids stay zero) through the C parser."""
    import pandas as pd
    with open(file_name) as fh:
        n, d = (int(t) for t in fh.readline().split())
    X = np.zeros((n, d))
    try:
        df = pd.read_csv(file_name, sep=r'\s+', header=None, skiprows=1, engine='c')
    except pd.errors.EmptyDataError:
        return X
    if df.shape[1] != d + 1:
        raise ValueError('%s: header says d=%d, rows have %d values' % (file_name, d, df.shape[1] - 1))
    X[df[0].to_numpy(dtype=np.int64)] = df.iloc[:, 1:].to_numpy(dtype=np.float64)
    return X


_EMB_MAGIC = b'GEMEMB01'


def saveEmbeddingBinary(X, file_name):
    """Binary container: 8-byte magic, int64 n, int64 d, 1-byte itemsize (4|8), 7 pad bytes, then n*d row-major little-endian floats."""
    X = np.ascontiguousarray(X)
    if X.ndim != 2 or X.dtype not in (np.float32, np.float64):
        raise ValueError('saveEmbeddingBinary: 2-D float32/float64 array expected')
    with open(file_name, 'wb') as fh:
        fh.write(_EMB_MAGIC)
        fh.write(np.array(X.shape, dtype='<i8').tobytes())
        fh.write(bytes([X.dtype.itemsize]) + b'\0' * 7)
        fh.write(X.astype(X.dtype.newbyteorder('<'), copy=False).tobytes())


def loadEmbeddingBinary(file_name, mmap=False):
    with open(file_name, 'rb') as fh:
        head = fh.read(32)
    if len(head) < 32 or head[:8] != _EMB_MAGIC:
        raise ValueError('%s: not a GEM-HIP binary embedding' % file_name)
    n, d = (int(v) for v in np.frombuffer(head[8:24], dtype='<i8'))
    dt = {4: '<f4', 8: '<f8'}.get(head[24])
    if dt is None or n < 0 or d < 0:
        raise ValueError('%s: corrupt header' % file_name)
    if mmap:
        return np.memmap(file_name, dtype=dt, mode='r', offset=32, shape=(n, d))
    X = np.fromfile(file_name, dtype=dt, offset=32)
    if X.size != n * d:
        raise ValueError('%s: truncated (%d of %d values)' % (file_name, X.size, n * d))
    return X.reshape(n, d)
