"""CPU oracle for the OCT segmentation hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product path
(``oct_segmentation_amd``) never does and fails loudly when the HIP library
is missing.

PARITY UNPINNED: the reference repository ships no tests, golden vectors or
checkpoints for this path, and the arithmetic itself lives in un-vendored
third-party packages (segmentation_models_pytorch==0.3.3, torchvision ResNet,
torch==2.2.2 -- reference ``environment.yaml:32,39``) that are not installed
here.  This package restates their published algorithms in plain torch CPU
ops (fp32), anchored on the reference's own call sites:

* ``src/models/smp/model.py:38-44``  smp.create_model(arch, encoder, in_channels, classes)
* ``src/models/smp/model.py:49-51``  get_preprocessing_params -> mean/std buffers
* ``src/models/smp/model.py:55``     DiceLoss(MULTILABEL_MODE, from_logits=True)
* ``src/models/smp/model.py:65-71``  forward = (x - mean) / std -> net
* ``src/models/smp/model.py:183-200`` predict (no normalisation, sigmoid > 0.5)
* ``src/models/smp/utils.py:13-36``  get_metrics (tp/fp/fn/tn -> iou, dice, f1, p, r)

Self-checks that stand in for golden vectors: conv-parameter counts equal the
known smp model sizes, state_dict key names follow the smp/torchvision module
tree, Dice/metric closed-form known answers (tests/test_oracle.py).
"""
from .nets import create_model, get_preprocessing_params, ENCODER_CHANNELS  # noqa: F401
from .losses import DiceBCELoss, DiceLoss, bce_with_logits, soft_dice_score  # noqa: F401
from .metrics import get_stats, iou_score, f1_score, precision, sensitivity, get_metrics  # noqa: F401
from .model import OracleOCTSegmentationModel  # noqa: F401
