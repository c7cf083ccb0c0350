from .ctc_models import EncDecCTCModel, EncDecCTCModelBPE, conformer_ctc_config  # noqa: F401
