"""Mirror of ``softgroup.ops`` (reference: softgroup/ops/__init__.py star-imports functions.py)."""
from .functions import *  # noqa: F401,F403
from .functions import (ball_query, ballquery_batch_p, bfs_cluster, bfs_cluster_segments,  # noqa: F401
                        get_mask_iou_on_cluster, get_mask_iou_on_pred, get_mask_label,
                        global_avg_pool, octree_ball_query, sec_max, sec_mean, sec_min,
                        voxelization, voxelization_idx)
