from gem_amd.utils.graph_util import *  # noqa: F401,F403
from gem_amd.utils.graph_util import saveGraphToEdgeListTxt, saveGraphToEdgeListTxtn2v, loadGraphFromEdgeListTxt, loadEmbedding  # noqa: F401
