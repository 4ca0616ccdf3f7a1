"""Drop-in `libs` package: put this repo's root FIRST on PYTHONPATH in front of the reference checkout and
`from libs import utils, pvlt` (reference main_vl.py:25) resolves `pvlt` / `vl_heads` here (MI355X HIP implementation)
while `utils`, `vl_scores`, ... still come from the reference's own libs/ directory (namespace extension below)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
