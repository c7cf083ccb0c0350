from .ctc_models import EncDecCTCModel, conformer_ctc_config  # noqa: F401
