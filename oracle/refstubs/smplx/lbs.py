from oracle.smplx_lbs import (vertices2joints, blend_shapes, batch_rigid_transform, batch_rodrigues, lbs)  # noqa: F401
