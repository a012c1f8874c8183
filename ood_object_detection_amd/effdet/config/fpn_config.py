"""BiFPN graph description.

Restates the node table the reference derives in effdet/config/fpn_config.py:6-38
(`bifpn_config`) and the name lookup at :172-184.  Only the BiFPN family is on
the hot path; PAN / QuFPN variants are out of scope.

For levels ``min_level..max_level`` the inputs are nodes ``0..L-1``.  A top-down
sweep adds one node per level from ``max_level-1`` down to ``min_level``
(inputs: the level's latest node and the latest node of the level above), then
a bottom-up sweep adds one node per level from ``min_level+1`` to ``max_level``
(inputs: every node of that level plus the latest node of the level below).
"""
from .config_utils import Config


def bifpn_config(min_level, max_level, weight_method=None):
    weight_method = weight_method or 'fastattn'
    num_levels = max_level - min_level + 1
    ids = {min_level + i: [i] for i in range(num_levels)}
    next_id = num_levels
    nodes = []
    for lvl in range(max_level - 1, min_level - 1, -1):
        nodes.append(Config(reduction=1 << lvl,
                            inputs_offsets=[ids[lvl][-1], ids[lvl + 1][-1]],
                            weight_method=weight_method))
        ids[lvl].append(next_id)
        next_id += 1
    for lvl in range(min_level + 1, max_level + 1):
        nodes.append(Config(reduction=1 << lvl,
                            inputs_offsets=list(ids[lvl]) + [ids[lvl - 1][-1]],
                            weight_method=weight_method))
        ids[lvl].append(next_id)
        next_id += 1
    p = Config()
    dict.__setitem__(p, 'nodes', nodes)
    return p


_WEIGHT_METHODS = {'bifpn_sum': 'sum', 'bifpn_attn': 'attn', 'bifpn_fa': 'fastattn'}


def get_fpn_config(fpn_name, min_level=3, max_level=7):
    if not fpn_name:
        fpn_name = 'bifpn_fa'
    if fpn_name not in _WEIGHT_METHODS:
        # reference raises KeyError for unknown names (fpn_config.py:184); the PAN/QuFPN
        # names it also knows are outside this build's scope and fail the same way.
        raise KeyError(fpn_name)
    return bifpn_config(min_level, max_level, _WEIGHT_METHODS[fpn_name])
