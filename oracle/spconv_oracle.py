"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's vendored sparse convolution (3-D).

Rulebook: follows mmdet3d/ops/spconv/include/spconv/geometry.h — getValidOutPos :24-84 (regular: the output
positions `val` with in = val * stride - padding + koff * dilation, offset = row-major (kz, ky, kx)),
getValidOutPosTranspose :86-151, getIndicePairsConv :153-199, getIndicePairsDeConv :201-247, getIndicePairsSubM
:249-298 — and the host wrapper spconv_ops.h:26-150 (submanifold: stride 1, padding = ksize // 2 forced :74-77).
Output rows of a regular / transposed convolution are numbered by ascending (b, z, y, x): the order of the reference's
GPU path (torch::_unique of the linear indices, spconv_ops.h:128); its CPU path numbers them by first appearance, so a
comparison with the compiled CPU templates goes through the coordinates.  Pairs of an offset are listed by ascending
input row (the CPU path's order).
PINNED: against those CPU templates compiled from the header where it lies (oracle/build_ref.build_spconv_rulebook,
binding oracle/ref_spconv_rulebook_binding.cpp) — live in tests/test_oracle.py and through tests/golden/spconv.npz.
Convolution arithmetic: indiceConv / indiceConvBackward (spconv_ops.h:256-446), i.e. per offset
out[pairs[k][1]] += in[pairs[k][0]] @ W[k]; evaluated in float64 here.  No reference-produced numbers exist for it
(the routine is CUDA / extension code that cannot be built here): the float part is pinned by definition only."""
import numpy as np


def conv_output_size(input_size, ksize, stride, padding, dilation):
    """ops.py:20-30"""
    return [(input_size[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) // stride[i] + 1 for i in range(3)]


def deconv_output_size(input_size, ksize, stride, padding, dilation, output_padding):
    """ops.py:33-43"""
    return [(input_size[i] - 1) * stride[i] - 2 * padding[i] + ksize[i] + output_padding[i] for i in range(3)]


def _candidates(indices, k_off, ksize, stride, padding, dilation, out_shape, transpose):
    """for one kernel offset: (valid [N] bool, out positions [N, 3])"""
    pos = indices[:, 1:4].astype(np.int64)
    ko = np.array(k_off, dtype=np.int64)
    st, pd, dl = (np.array(v, dtype=np.int64) for v in (stride, padding, dilation))
    if transpose:
        out = pos * st - pd + ko * dl
        valid = np.ones(len(pos), dtype=bool)
    else:
        num = pos + pd - ko * dl
        valid = ((num >= 0) & (num % st == 0)).all(1)
        out = num // st
    valid &= ((out >= 0) & (out < np.array(out_shape, dtype=np.int64))).all(1)
    return valid, out


def indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, out_padding=(0, 0, 0),
                 subm=False, transpose=False):
    """-> outids [M, 4] int32, pairs [K, 2, N] int32 (-1 filled), num [K] int32, out_shape"""
    indices = np.asarray(indices, dtype=np.int32)
    n = len(indices)
    if subm:
        stride, padding = [1, 1, 1], [k // 2 for k in ksize]
        out_shape = list(spatial_shape)
    elif transpose:
        out_shape = deconv_output_size(spatial_shape, ksize, stride, padding, dilation, out_padding)
    else:
        out_shape = conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    kvol = int(np.prod(ksize))
    vol = int(np.prod(out_shape))

    def lin(b, p):
        return ((b.astype(np.int64) * out_shape[0] + p[:, 0]) * out_shape[1] + p[:, 1]) * out_shape[2] + p[:, 2]

    cands = []
    for k in range(kvol):
        k_off = (k // (ksize[1] * ksize[2]), (k // ksize[2]) % ksize[1], k % ksize[2])
        valid, out = _candidates(indices, k_off, ksize, stride, padding, dilation, out_shape, transpose)
        cands.append((valid, lin(indices[:, 0], out)))
    if subm:
        out_lin = lin(indices[:, 0], indices[:, 1:4].astype(np.int64))
        order = np.argsort(out_lin, kind='stable')
        sorted_lin = out_lin[order]
        outids = indices.copy()
    else:
        all_lin = np.concatenate([c[1][c[0]] for c in cands]) if n else np.zeros(0, dtype=np.int64)
        sorted_lin = np.unique(all_lin)
        order = np.arange(len(sorted_lin))
        b = sorted_lin // vol
        rem = sorted_lin % vol
        outids = np.stack([b, rem // (out_shape[1] * out_shape[2]), (rem // out_shape[2]) % out_shape[1],
                           rem % out_shape[2]], 1).astype(np.int32)
    pairs = np.full((kvol, 2, n), -1, dtype=np.int32)
    num = np.zeros(kvol, dtype=np.int32)
    for k, (valid, l) in enumerate(cands):
        if len(sorted_lin) == 0:
            continue
        pos = np.searchsorted(sorted_lin, l)
        pos_c = np.minimum(pos, len(sorted_lin) - 1)
        hit = valid & (sorted_lin[pos_c] == l)
        j = np.nonzero(hit)[0]
        num[k] = len(j)
        pairs[k, 0, :len(j)] = j
        pairs[k, 1, :len(j)] = order[pos_c[j]]
    return outids, pairs, num, out_shape


def maps_from_pairs(pairs, num, n_in, n_out):
    kvol = pairs.shape[0]
    in2out = np.full((kvol, n_in), -1, dtype=np.int32)
    out2in = np.full((kvol, n_out), -1, dtype=np.int32)
    for k in range(kvol):
        a, b = pairs[k, 0, :num[k]], pairs[k, 1, :num[k]]
        in2out[k, a] = b
        out2in[k, b] = a
    return in2out, out2in


def indice_conv(features, filters, pairs, num, n_out, inverse=False):
    """spconv_ops.h:256-357 in float64; filters [kz, ky, kx, Cin, Cout]"""
    x = np.asarray(features, dtype=np.float64)
    w = np.asarray(filters, dtype=np.float64).reshape(-1, filters.shape[-2], filters.shape[-1])
    out = np.zeros((n_out, w.shape[2]))
    for k in range(w.shape[0]):
        src, dst = pairs[k, int(inverse), :num[k]], pairs[k, 1 - int(inverse), :num[k]]
        np.add.at(out, dst, x[src] @ w[k])
    return out


def indice_conv_backward(features, filters, out_grad, pairs, num, inverse=False):
    """spconv_ops.h:359-446 -> (input gradient, filter gradient)"""
    x = np.asarray(features, dtype=np.float64)
    g = np.asarray(out_grad, dtype=np.float64)
    w = np.asarray(filters, dtype=np.float64).reshape(-1, filters.shape[-2], filters.shape[-1])
    dx = np.zeros_like(x)
    dw = np.zeros_like(w)
    for k in range(w.shape[0]):
        src, dst = pairs[k, int(inverse), :num[k]], pairs[k, 1 - int(inverse), :num[k]]
        dw[k] = x[src].T @ g[dst]
        np.add.at(dx, src, g[dst] @ w[k].T)
    return dx, dw.reshape(filters.shape)


def indice_maxpool(features, pairs, num, n_out):
    """pool_ops.h:24-58 + maxpool.cc:22-40: output starts at ZERO, then out = max(out, in) per pair"""
    x = np.asarray(features, dtype=np.float32)
    out = np.zeros((n_out, x.shape[1]), dtype=np.float32)
    for k in range(pairs.shape[0]):
        src, dst = pairs[k, 0, :num[k]], pairs[k, 1, :num[k]]
        np.maximum.at(out, dst, x[src])
    return out


def indice_maxpool_backward(features, out_features, out_grad, pairs, num):
    """pool_ops.h:60-96 + maxpool.cc:42-63: every input equal to the pooled value receives the gradient"""
    x = np.asarray(features, dtype=np.float32)
    y = np.asarray(out_features, dtype=np.float32)
    g = np.asarray(out_grad, dtype=np.float32)
    dx = np.zeros_like(x)
    for k in range(pairs.shape[0]):
        src, dst = pairs[k, 0, :num[k]], pairs[k, 1, :num[k]]
        np.add.at(dx, src, np.where(y[dst] == x[src], g[dst], np.float32(0)))
    return dx
