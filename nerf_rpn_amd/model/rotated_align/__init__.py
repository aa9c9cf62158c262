from .roi_align_rotate_3d import ROIAlignRotated3D, roi_align_rotated_3d  # noqa: F401
