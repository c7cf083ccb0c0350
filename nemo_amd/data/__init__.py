"""Input side of the training hot path: manifests, map-style speech datasets, the
padding collate function, duration-aware batch shaping across ranks, and the pinned-memory hand-over to HBM."""
from .audio import load_audio
from .audio_to_text import AudioToBPEDataset, AudioToCharDataset, _speech_collate_fn
from .batching import DurationBucketBatchSampler, SemiSortBatchSampler
from .loader import DeviceBatchLoader
from .manifest import ASRAudioText, AudioTextEntity, item_iter, parse_item
from .text import CharParser, SentencePieceTokenizer, TokenizerWrapper, make_parser

__all__ = ["load_audio", "AudioToBPEDataset", "AudioToCharDataset", "_speech_collate_fn", "SemiSortBatchSampler", "DurationBucketBatchSampler",
           "DeviceBatchLoader", "ASRAudioText", "AudioTextEntity", "item_iter", "parse_item", "CharParser",
           "SentencePieceTokenizer", "TokenizerWrapper", "make_parser"]
