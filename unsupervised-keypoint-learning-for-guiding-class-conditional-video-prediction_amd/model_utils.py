"""Math helpers with the reference's names (reference: utils/model.py)."""
from . import ops
from .variables import Sym, is_sym


def get_coords_xy(x):
    """Both get_coord calls + tf.stack of pose_encoder in one fused pass (reference networks/__init__.py:68-71).
    x [B,H,W,K] -> (mu [B,K,2] as (x,y), prob_y [B,H,K], prob_x [B,W,K])."""
    if is_sym(x):
        b, h, w, k = x.shape
        return Sym(b, k, 2), Sym(b, h, k), Sym(b, w, k)
    return ops.keypoint_head(x)


def get_coord(x, other_axis, axis_size):
    """reference get_coord (utils/model.py:63-70): returns (coordinate [B,K], probability [B,axis_size,K]).
    other_axis=2 -> y (softmax over H), other_axis=1 -> x (softmax over W)."""
    mu, prob_y, prob_x = get_coords_xy(x)
    if other_axis == 2:
        assert axis_size == x.shape[1]
        return mu[:, :, 1], prob_y
    assert other_axis == 1 and axis_size == x.shape[2]
    return mu[:, :, 0], prob_x


def get_gaussian_maps(mu, shape_hw, inv_std=14.3):
    """reference get_gaussian_maps (utils/model.py:49-60): mu [B,K,2] (x,y) -> [B,H,W,K]."""
    if is_sym(mu):
        return Sym(mu.shape[0], shape_hw[0], shape_hw[1], mu.shape[1])
    return ops.gaussian_maps(mu, int(shape_hw[0]), int(shape_hw[1]), inv_std)
