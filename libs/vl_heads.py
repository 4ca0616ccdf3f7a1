"""`libs.vl_heads` -- the head parameter holders of the MI355X model under the reference's class names
(libs/vl_heads.py: MLMHead, ITMHead, CLSHead, ITGHead).  They carry parameters only; the kernels that evaluate them
are scheduled by mvlt_amd.schedule."""
from mvlt_amd.pvlt import _ClsHead as CLSHead  # noqa: F401
from mvlt_amd.pvlt import _ClsHead as ITMHead  # noqa: F401
from mvlt_amd.pvlt import _ITGHead as ITGHead  # noqa: F401
from mvlt_amd.pvlt import _MLMHead as MLMHead  # noqa: F401
