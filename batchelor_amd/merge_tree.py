"""Host-side merge-tree handling: mirrors R/MNN_tree.R:21-46 (.binarize_tree) and :80-107 (the leaf checks of
.create_tree_predefined) and flattens the binary tree into the post-order encoding bmx_engine_run() takes.

R lists are Python lists / tuples; leaves are 1-based batch numbers or batch names.  A flat sequence (R's
`merge.order=c(3,1,2)`) is a progressive merge.
"""
from __future__ import annotations

import numpy as np


def _is_leaf(x):
    return not isinstance(x, (list, tuple))


def _as_list(x):
    return [v.item() if hasattr(v, "item") else v for v in x] if isinstance(x, np.ndarray) else x


def binarize_tree(tree):
    """R/MNN_tree.R:21-46: progressive merge for > 2 children, single-child nodes collapsed."""
    tree = _as_list(tree)
    if _is_leaf(tree):
        return tree
    n = len(tree)
    if n == 0:
        raise ValueError("merge tree contains a node with no children")
    if n == 1:
        return binarize_tree(tree[0])
    cur = [binarize_tree(tree[0]), binarize_tree(tree[1])]
    for child in tree[2:]:
        cur = [cur, binarize_tree(child)]
    return cur


def tree_leaves(tree):
    if _is_leaf(tree):
        return [tree]
    return [leaf for child in tree for leaf in tree_leaves(child)]


def resolve_merge_order(nbatches, merge_order=None, names=None):
    """The binary tree with 1-based integer leaves (R/MNN_tree.R:80-107); raises the reference's error text."""
    if merge_order is None:
        merge_order = list(range(1, nbatches + 1))
    merge_order = _as_list(merge_order)
    tree = binarize_tree(merge_order)
    leaves = tree_leaves(tree)
    numeric = all(isinstance(x, (int, float, np.integer, np.floating)) and not isinstance(x, bool) for x in leaves)
    if numeric:
        resolved = [int(x) for x in leaves]
    else:
        lookup = {str(nm): i + 1 for i, nm in enumerate(names or [])}
        resolved = [lookup.get(str(x)) for x in leaves]
    bad = (any(x is None for x in resolved) or len(set(resolved)) != len(resolved)
           or any(x < 1 or x > nbatches for x in resolved) or len(resolved) != nbatches)
    if bad:
        raise ValueError("invalid leaf nodes specified in 'merge.order'")
    it = iter(resolved)

    def relist(t):
        return next(it) if _is_leaf(t) else [relist(c) for c in t]

    return relist(tree)


def encode_postorder(tree):
    """Binary tree -> int32 post-order code: leaf = batch id, 0 = merge (left child is the deeper stack entry)."""
    out = []

    def walk(t):
        if _is_leaf(t):
            out.append(int(t))
            return
        if len(t) != 2:
            raise ValueError("merge tree structure should contain two children per node")
        walk(t[0])
        walk(t[1])
        out.append(0)

    walk(tree)
    return np.asarray(out, dtype=np.int32)
