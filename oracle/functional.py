"""TEST INFRASTRUCTURE — CPU oracle: a functional restatement (plain PyTorch, fp32 or fp64, no classes, no
state) of the reference's hot-path algorithms, each citing the reference file:line it follows.  Parameters are
passed as a dict keyed by the reference's own state_dict names so the same weights drive the reference module,
this oracle and the HIP modules.

Pinned: tests/test_oracle_vs_reference.py runs these functions against the reference's actual module files
(oracle/ref_import.py) wherever /root/reference exists, and against tests/golden/*.npz everywhere.
The PyG primitives underneath (oracle/pyg_restated.py) are unpinned — see that file's header.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product never does.
"""
import torch

from . import pyg_restated as P


def propagate_add(edge_index, x, norm):
    """MessagePassing.propagate(aggr="add", flow="source_to_target", node_dim=-2) with message = norm * x_j
    (dcrnn.py:39-40; PyG semantics in SURVEY.md Appendix A): gather sources, scale, index_add_ into targets."""
    src, dst = edge_index[0], edge_index[1]
    x_j = x.index_select(-2, src)
    if norm is not None:
        m = (norm.view(-1, 1) if norm.dim() == 1 else norm.unsqueeze(-1)) * x_j
    else:
        m = x_j
    out = x.new_zeros(x.shape)
    return out.index_add_(x.dim() - 2, dst, m)


# ------------------------------------------------------------------------------------------------ DCRNN family

def _diffusion_terms(X, edge_index, norm_out, reverse_edge_index, norm_in, weight, bias):
    """dcrnn.py:79-111 (identical at :292-325): H = sum_k T_k^o W[0,k] + T_k^i W[1,k] + b with
    T_1 = P X and T_k = 2 P T_{k-1} - Tx_0 where Tx_0 stays X forever (dcrnn.py:106: `Tx_0 ... = Tx_1 ...`)."""
    K = weight.size(1)
    Tx_0 = X
    H = X @ weight[0, 0] + X @ weight[1, 0]
    if K > 1:
        T_o = propagate_add(edge_index, X, norm_out)
        T_i = propagate_add(reverse_edge_index, X, norm_in)
        H = H + T_o @ weight[0, 1] + T_i @ weight[1, 1]
    for k in range(2, K):
        T_o = 2.0 * propagate_add(edge_index, T_o, norm_out) - Tx_0
        T_i = 2.0 * propagate_add(reverse_edge_index, T_i, norm_in) - Tx_0
        H = H + T_o @ weight[0, k] + T_i @ weight[1, k]
    if bias is not None:
        H = H + bias
    return H


def dconv(X, edge_index, edge_weight, weight, bias):
    """DConv.forward, dense-adjacency path (dcrnn.py:59-77): N' x N' adjacency (duplicates summed), degrees as
    row/col sums, reciprocals, norm_out = 1/deg_out[row], norm_in = 1/deg_in[row] (sic), reverse list =
    dense_to_sparse(adj^T) (row-major nonzeros of the transpose, exact zeros dropped)."""
    adj = P.to_dense_adj(edge_index, edge_attr=edge_weight)
    adj = adj.reshape(adj.size(1), adj.size(2)).to(X.dtype)
    ones = torch.ones(adj.size(0), 1, dtype=X.dtype)
    deg_out = (adj @ ones).flatten()
    deg_in = (ones.t() @ adj).flatten()
    row = edge_index[0]
    norm_out = torch.reciprocal(deg_out)[row]
    norm_in = torch.reciprocal(deg_in)[row]
    reverse_edge_index, _ = P.dense_to_sparse(adj.transpose(0, 1))
    return _diffusion_terms(X, edge_index, norm_out, reverse_edge_index, norm_in, weight, bias)


def dconv_norms_scatter(edge_index, edge_weight, num_nodes, dtype=torch.float32):
    """BatchedDConv.forward graph prep (dcrnn.py:277-290): scatter_add degrees, reciprocals, norm gathers by `row`,
    reverse list = [col,row] sorted by col * num_nodes + row."""
    row, col = edge_index[0], edge_index[1]
    w = edge_weight.to(dtype)
    deg_out = torch.zeros(num_nodes, dtype=dtype).scatter_add_(0, row, w)
    deg_in = torch.zeros(num_nodes, dtype=dtype).scatter_add_(0, col, w)
    norm_out = torch.reciprocal(deg_out)[row]
    norm_in = torch.reciprocal(deg_in)[row]
    rev = torch.stack([col, row], dim=0)
    order = (rev[0] * num_nodes + rev[1]).argsort(stable=True)
    return norm_out, norm_in, rev[:, order]


def batched_dconv(X, edge_index, edge_weight, weight, bias):
    """BatchedDConv.forward (dcrnn.py:258-325), O(E) scatter form."""
    norm_out, norm_in, rev = dconv_norms_scatter(edge_index, edge_weight, X.size(0), X.dtype)
    return _diffusion_terms(X, edge_index, norm_out, rev, norm_in, weight, bias)


def _gru(conv, X, H, p):
    """GRU gate chain shared by DCRNN (dcrnn.py:172-192) and BatchedDCRNN (:406-427)."""
    XH = torch.cat([X, H], dim=1)
    Z = torch.sigmoid(conv(XH, p["conv_x_z.weight"], p.get("conv_x_z.bias")))
    R = torch.sigmoid(conv(XH, p["conv_x_r.weight"], p.get("conv_x_r.bias")))
    Ht = torch.tanh(conv(torch.cat([X, H * R], dim=1), p["conv_x_h.weight"], p.get("conv_x_h.bias")))
    return Z * H + (1 - Z) * Ht


def dcrnn_cell(X, edge_index, edge_weight, H, p):
    """DCRNN.forward (dcrnn.py:194-219); H None -> zeros (:167-170)."""
    if H is None:
        H = torch.zeros(X.shape[0], p["conv_x_z.weight"].size(3), dtype=X.dtype)
    return _gru(lambda x, w, b: dconv(x, edge_index, edge_weight, w, b), X, H, p)


def replicate_edge_index(edge_index, batch_size, num_nodes):
    """BatchedDCRNN._replicate_edge_index (dcrnn.py:363-369): block-diagonal copies offset by b * num_nodes."""
    return torch.cat([edge_index + b * num_nodes for b in range(batch_size)], dim=1)


def batched_dcrnn(X, edge_index, edge_weight, p):
    """BatchedDCRNN.forward (dcrnn.py:429-475): X [B,T,N,F] -> [B,T,N,O]; hidden state starts at zero."""
    B, T, N, F = X.shape
    O = p["conv_x_z.weight"].size(3)
    ei = replicate_edge_index(edge_index, B, N)
    ew = edge_weight.repeat(B)
    H = torch.zeros(B * N, O, dtype=X.dtype)
    outs = []
    for t in range(T):
        x_t = X[:, t].reshape(B * N, F)
        H = _gru(lambda x, w, b: batched_dconv(x, ei, ew, w, b), x_t, H, p)
        outs.append(H.reshape(B, N, O))
    return torch.stack(outs, dim=1)


# ------------------------------------------------------------------------------------------------ GCN family

def gcn_conv(x, edge_index, edge_weight, lin_weight, bias, improved=False, add_self_loops=True):
    """PyG GCNConv.forward as used by temporalgcn.py:38-70 (2-D [N,F] or 3-D [B,N,F] input, node_dim=-2)."""
    ei, w = P.gcn_norm(edge_index, edge_weight, x.size(-2), improved, add_self_loops, dtype=x.dtype)
    x = x @ lin_weight.t()
    out = propagate_add(ei, x, w.to(x.dtype))
    if bias is not None:
        out = out + bias
    return out


def tgcn_cell(X, edge_index, edge_weight, H, p, improved=False, add_self_loops=True):
    """TGCN.forward (temporalgcn.py:104-130) and TGCN2.forward (:187-233): gates :82-102 / :205-226."""
    O = p["linear_z.weight"].size(0)
    if H is None:
        H = torch.zeros(*X.shape[:-1], O, dtype=X.dtype)

    def gate(name, h_in):
        conv = gcn_conv(X, edge_index, edge_weight, p[f"conv_{name}.lin.weight"], p[f"conv_{name}.bias"],
                        improved, add_self_loops)
        cat = torch.cat([conv, h_in], dim=-1)
        return cat @ p[f"linear_{name}.weight"].t() + p[f"linear_{name}.bias"]

    Z = torch.sigmoid(gate("z", H))
    R = torch.sigmoid(gate("r", H))
    Ht = torch.tanh(gate("h", H * R))
    return Z * H + (1 - Z) * Ht


def a3tgcn(X, edge_index, edge_weight, H, p, improved=False, add_self_loops=True):
    """A3TGCN.forward (attentiontemporalgcn.py:52-79) / A3TGCN2.forward (:130-157): X [..., F, P]; every period
    uses the same H; output = sum_p softmax(attention)_p * TGCN(X[..., p], H)."""
    probs = torch.softmax(p["_attention"].to(X.dtype), dim=0)
    base = {k[len("_base_tgcn."):]: v for k, v in p.items() if k.startswith("_base_tgcn.")}
    acc = 0
    for period in range(probs.numel()):
        acc = acc + probs[period] * tgcn_cell(X[..., period], edge_index, edge_weight, H, base, improved,
                                              add_self_loops)
    return acc


# ------------------------------------------------------------------------------------------------ EvolveGCN-H

def gru_step(x, h, p, prefix="recurrent_layer."):
    """torch.nn.GRU(num_layers=1) on a length-1 sequence: one cell step with nn.GRU's gate order (r, z, n)."""
    wi, wh = p[prefix + "weight_ih_l0"], p[prefix + "weight_hh_l0"]
    bi, bh = p[prefix + "bias_ih_l0"], p[prefix + "bias_hh_l0"]
    gi, gh = x @ wi.t() + bi, h @ wh.t() + bh
    i_r, i_z, i_n = gi.chunk(3, dim=-1)
    h_r, h_z, h_n = gh.chunk(3, dim=-1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def evolvegcnh_step(X, edge_index, edge_weight, W, p, improved=False, add_self_loops=True):
    """One snapshot of EvolveGCNH.forward (evolvegcnh.py:78-102): the node summary from TopKPooling(ratio =
    in_channels / num_nodes) (:63, PyG: score = tanh(x.p / |p|), top-k rows scaled by their score) drives one GRU step
    on the [F, F] weight matrix (:95-100), which then transforms X before the normalised propagate of GCNConv_Fixed_W
    (evolvegcno.py:81-98).  W: the carried weight [F, F] (p["initial_weight"][0] at the first snapshot).
    Returns (out [N, F], new W [F, F])."""
    n, Fdim = X.shape
    w = p["pooling_layer.select.weight"]
    score = torch.tanh((X * w).sum(dim=-1) / w.norm(p=2, dim=-1))
    k = int((float(Fdim / n) * torch.tensor(n).to(score.dtype)).ceil().to(torch.long))
    perm = torch.sort(score.view(-1), descending=True).indices[:k]
    x_tilde = X[perm] * score[perm].view(-1, 1)                       # [F, F]: batch of F "sequences", input size F
    W_new = gru_step(x_tilde, W, p)
    ei, nw = P.gcn_norm(edge_index, edge_weight, n, improved, add_self_loops, dtype=X.dtype)
    out = propagate_add(ei, X @ W_new, nw.to(X.dtype))
    return out, W_new


# ------------------------------------------------------------------------------------------------ Chebyshev family

def cheb_norm(edge_index, edge_weight, num_nodes, normalization, lambda_max, dtype):
    """PyG ChebConv.__norm__ (scaled Laplacian 2L/lambda_max - I, "-1" folded into the diagonal entries)."""
    ei, w = P.get_laplacian(edge_index, edge_weight, normalization, dtype, num_nodes)
    if lambda_max is None:
        lambda_max = 2.0 * w.max()
    w = (2.0 * w) / lambda_max
    w.masked_fill_(w == float("inf"), 0)
    w[ei[0] == ei[1]] -= 1
    return ei, w


def cheb_conv(x, edge_index, edge_weight, lin_weights, bias, normalization="sym", lambda_max=None):
    """PyG ChebConv.forward as used by stgcn.py:115-121,151-153 and gconv_gru.py:57-107."""
    ei, norm = cheb_norm(edge_index, edge_weight, x.size(-2), normalization, lambda_max, x.dtype)
    Tx_0 = x
    Tx_1 = x
    out = Tx_0 @ lin_weights[0].t()
    if len(lin_weights) > 1:
        Tx_1 = propagate_add(ei, x, norm)
        out = out + Tx_1 @ lin_weights[1].t()
    for W in lin_weights[2:]:
        Tx_2 = 2.0 * propagate_add(ei, Tx_1, norm) - Tx_0
        out = out + Tx_2 @ W.t()
        Tx_0, Tx_1 = Tx_1, Tx_2
    if bias is not None:
        out = out + bias
    return out


def gconv_gru_cell(X, edge_index, edge_weight, H, p, K, normalization="sym", lambda_max=None):
    """GConvGRU.forward (gconv_gru.py:119-170): Z = s(cheb_xz(X) + cheb_hz(H)), R likewise (:121-133),
    H~ = tanh(cheb_xh(X) + cheb_hh(H * R)) (:135-139), H' = Z * H + (1 - Z) * H~ (:141-143)."""
    def conv(name, x):
        lins = [p[f"{name}.lins.{k}.weight"] for k in range(K)]
        return cheb_conv(x, edge_index, edge_weight, lins, p.get(f"{name}.bias"), normalization, lambda_max)
    if H is None:
        H = torch.zeros(X.size(0), p["conv_h_z.lins.0.weight"].size(0), dtype=X.dtype)
    Z = torch.sigmoid(conv("conv_x_z", X) + conv("conv_h_z", H))
    R = torch.sigmoid(conv("conv_x_r", X) + conv("conv_h_r", H))
    Ht = torch.tanh(conv("conv_x_h", X) + conv("conv_h_h", H * R))
    return Z * H + (1 - Z) * Ht


def cheb_conv_attention(x, edge_index, spatial_attention, edge_weight, weight, bias, normalization, lambda_max=None):
    """ChebConvAttention.forward (astgcn.py:112-183) with its in-tree __norm__ (:82-110): x [B,N,Fin],
    spatial_attention [B,N,N], weight [K,Fin,Fout]."""
    if normalization != "sym" and lambda_max is None:
        raise ValueError("You need to pass `lambda_max` to `forward() in`case the normalization is non-symmetric.")
    N = x.size(-2)
    lam = torch.tensor(2.0 if lambda_max is None else float(lambda_max), dtype=x.dtype)
    # __norm__ (:82-110)
    ei, w = P.remove_self_loops(edge_index, edge_weight)
    ei, w = P.get_laplacian(ei, w, normalization, x.dtype, N)
    w = (2.0 * w) / lam
    w = w.masked_fill(w == float("inf"), 0)
    ei, w = P.add_self_loops(ei, w, fill_value=-1.0, num_nodes=N)
    row, col = ei
    att_norm = w * spatial_attention[:, row, col]                                     # :157  [B, E']
    eye = torch.eye(N, dtype=x.dtype)
    TAx_0 = torch.matmul((eye * spatial_attention).permute(0, 2, 1), x)               # :159-164
    out = torch.matmul(TAx_0, weight[0])
    eit = ei[[1, 0]]                                                                  # :166
    TAx_1 = TAx_0
    if weight.size(0) > 1:
        # propagate(edge_index_transpose, x=TAx_0, norm=Att_norm): out[:, eit[1]] += Att_norm[:, e, None] * x[:, eit[0]]
        msg = att_norm.unsqueeze(-1) * TAx_0[:, eit[0]]
        TAx_1 = torch.zeros_like(TAx_0).index_add_(1, eit[1], msg)
        out = out + torch.matmul(TAx_1, weight[1])
    for k in range(2, weight.size(0)):
        TAx_2 = propagate_add(eit, TAx_1, w)
        TAx_2 = 2.0 * TAx_2 - TAx_0
        out = out + torch.matmul(TAx_2, weight[k])
        TAx_0, TAx_1 = TAx_1, TAx_2
    if bias is not None:
        out = out + bias
    return out
