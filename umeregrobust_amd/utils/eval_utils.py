"""Drop-in counterparts of the hot-path symbols of the reference's utils/eval_utils.py."""
from .. import ops


def relative_rotation_error(R, R_hat):
    """reference utils/eval_utils.py:60-76: rotation error in degrees, [b,3,3] x2 -> [b]."""
    return ops.rre_deg(R, R_hat)
