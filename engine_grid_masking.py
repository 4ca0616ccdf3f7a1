"""`engine_grid_masking` -- same import path and names as the reference module (engine_grid_masking.py), so
reference main_vl.py:198 (`from engine_grid_masking import evaluate_vl, train_one_epoch_vl, visual_vl,
evaluate_retrieval, evaluate_recognition`) imports unchanged.  Implementation: mvlt_amd/engine.py, mvlt_amd/evaluate.py."""
from mvlt_amd.engine import (ITM_LOSS_WEIGHT, MLM_LOSS_WEIGHT, T2I_LOSS_WEIGHT, train_one_epoch,  # noqa: F401
                             train_one_epoch_vl)
from mvlt_amd.evaluate import evaluate_recognition, evaluate_retrieval, evaluate_vl, visual_vl  # noqa: F401
