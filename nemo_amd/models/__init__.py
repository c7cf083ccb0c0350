from .ctc_models import (EncDecCTCModel, EncDecCTCModelBPE, conformer_ctc_config,  # noqa: F401
                         squeezeformer_ctc_config)
from .rnnt_models import EncDecRNNTModel, fastconformer_transducer_config  # noqa: F401
