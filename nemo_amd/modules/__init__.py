from .audio_preprocessing import (AudioToMelSpectrogramPreprocessor, FilterbankFeatures,  # noqa: F401
                                  SpectrogramAugmentation)
from .conformer_encoder import ConformerEncoder  # noqa: F401
from .squeezeformer_encoder import SqueezeformerEncoder  # noqa: F401
from .conv_asr import ConvASRDecoder  # noqa: F401
from .ctc import CTCLoss  # noqa: F401
from .ctc_decoding import GreedyCTCDecoder, WER, word_error_rate  # noqa: F401
from .rnnt_loss import RNNTLoss, RNNTLossNumba  # noqa: F401
from .rnnt import RNNTDecoder, RNNTJoint  # noqa: F401
from .rnnt_decoding import GreedyBatchedRNNTInfer, RNNTDecoding, RNNTWER, Hypothesis  # noqa: F401
