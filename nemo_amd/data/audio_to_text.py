"""Map-style speech datasets and their collate function: the input side of `EncDecCTCModel.training_step`, whose batch is
`(signal [B, S] f32, signal_len [B] i64, transcript [B, U] i64, transcript_len [B] i64)` (ctc_models.py:549-555).

Mirrors `nemo/collections/asr/data/audio_to_text.py`:
  * `_speech_collate_fn` (:52-110): pad every signal with zeros to the longest of the batch, every token list with
    `pad_id`, stack; an optional 5th field (sample ids) becomes an int32 tensor;
  * `ASRManifestProcessor` (:113-180): bos / eos wrapping in `process_text_by_sample`;
  * `_AudioTextDataset` (:392-509), `AudioToCharDataset` (:512-607), `AudioToBPEDataset` (:610-718): the bos / eos / pad
    rules of the BPE variant (ids used only when > 0, pad defaults to 0).
"""
from __future__ import annotations

from collections.abc import Iterable
from typing import Callable, List, Optional, Union

import torch
from torch.utils.data import Dataset

from .audio import load_audio
from .manifest import ASRAudioText
from .text import TokenizerWrapper, make_parser


def _speech_collate_fn(batch, pad_id):
    packed = list(zip(*batch))
    if len(packed) == 5:
        _, audio_lengths, _, tokens_lengths, sample_ids = packed
    elif len(packed) == 4:
        sample_ids = None
        _, audio_lengths, _, tokens_lengths = packed
    else:
        raise ValueError("Expects 4 or 5 tensors in the batch!")
    has_audio = audio_lengths[0] is not None
    has_tokens = tokens_lengths[0] is not None
    B = len(batch)
    audio_signal = audio_lens = tokens = tokens_lens = None
    if has_audio:
        audio_lens = torch.stack(list(audio_lengths))
        max_audio = int(audio_lens.max())
        # one zero-filled buffer, one copy per utterance (the reference pads each signal and stacks: two copies)
        audio_signal = batch[0][0].new_zeros((B, max_audio))
        for i, b in enumerate(batch):
            n = int(b[1])
            audio_signal[i, :n] = b[0][:n]
    if has_tokens:
        tokens_lens = torch.stack(list(tokens_lengths))
        max_tok = int(tokens_lens.max())
        tokens = batch[0][2].new_full((B, max_tok), pad_id)
        for i, b in enumerate(batch):
            n = int(b[3])
            tokens[i, :n] = b[2][:n]
    if sample_ids is None:
        return audio_signal, audio_lens, tokens, tokens_lens
    return audio_signal, audio_lens, tokens, tokens_lens, torch.tensor(sample_ids, dtype=torch.int32)


class ASRManifestProcessor:
    def __init__(self, manifest_filepath: str, parser: Union[str, Callable], max_duration: Optional[float] = None,
                 min_duration: Optional[float] = None, max_utts: int = 0, bos_id: Optional[int] = None,
                 eos_id: Optional[int] = None, pad_id: int = 0, manifest_parse_func: Optional[Callable] = None):
        self.parser = parser
        self.collection = ASRAudioText(manifest_filepath, parser=parser, min_duration=min_duration, max_duration=max_duration,
                                       max_number=max_utts, parse_func=manifest_parse_func)
        self.eos_id, self.bos_id, self.pad_id = eos_id, bos_id, pad_id

    def process_text_by_sample(self, sample):
        t, tl = list(sample.text_tokens), len(sample.text_tokens)
        if self.bos_id is not None:
            t = [self.bos_id] + t
            tl += 1
        if self.eos_id is not None:
            t = t + [self.eos_id]
            tl += 1
        return t, tl


class _AudioTextDataset(Dataset):
    def __init__(self, manifest_filepath: str, parser: Union[str, Callable], sample_rate: int, int_values: bool = False,
                 augmentor=None, max_duration: Optional[float] = None, min_duration: Optional[float] = None, max_utts: int = 0,
                 trim: bool = False, bos_id: Optional[int] = None, eos_id: Optional[int] = None, pad_id: int = 0,
                 return_sample_id: bool = False, channel_selector=None, manifest_parse_func: Optional[Callable] = None):
        if augmentor is not None:
            raise NotImplementedError("waveform perturbation (AudioAugmentor) is outside the training hot path built here")
        if trim:
            raise NotImplementedError("silence trimming needs librosa.effects.trim (not in this image)")
        if not isinstance(manifest_filepath, str):
            manifest_filepath = ",".join(manifest_filepath)
        self.manifest_processor = ASRManifestProcessor(manifest_filepath, parser, max_duration, min_duration, max_utts,
                                                       bos_id, eos_id, pad_id, manifest_parse_func)
        self.sample_rate, self.int_values = sample_rate, int_values
        self.return_sample_id, self.channel_selector = return_sample_id, channel_selector

    def get_manifest_sample(self, sample_id):
        return self.manifest_processor.collection[sample_id]

    def __getitem__(self, index):
        if isinstance(index, Iterable):
            return [self._process_sample(int(i)) for i in index]
        return self._process_sample(index)

    def _process_sample(self, index):
        sample = self.manifest_processor.collection[index]
        f = load_audio(sample.audio_file, self.sample_rate, offset=sample.offset or 0, duration=sample.duration,
                       int_values=self.int_values, channel_selector=self.channel_selector)
        fl = torch.tensor(f.shape[0]).long()
        t, tl = self.manifest_processor.process_text_by_sample(sample)
        out = f, fl, torch.tensor(t).long(), torch.tensor(tl).long()
        return out + (index,) if self.return_sample_id else out

    def __len__(self):
        return len(self.manifest_processor.collection)

    def _collate_fn(self, batch):
        if batch and isinstance(batch[0], list):  # a batch sampler used as `sampler` hands over one list of samples
            batch = batch[0]
        return _speech_collate_fn(batch, pad_id=self.manifest_processor.pad_id)

    @property
    def durations(self) -> List[float]:
        return self.manifest_processor.collection.durations


class AudioToCharDataset(_AudioTextDataset):
    def __init__(self, manifest_filepath: str, labels: List[str], sample_rate: int, int_values: bool = False, augmentor=None,
                 max_duration: Optional[float] = None, min_duration: Optional[float] = None, max_utts: int = 0,
                 blank_index: int = -1, unk_index: int = -1, normalize: bool = True, trim: bool = False,
                 bos_id: Optional[int] = None, eos_id: Optional[int] = None, pad_id: int = 0, parser: Union[str, Callable] = "base",
                 return_sample_id: bool = False, channel_selector=None, manifest_parse_func: Optional[Callable] = None):
        self.labels = labels
        if not callable(parser):
            parser = make_parser(labels=labels, name=parser, unk_id=unk_index, blank_id=blank_index, do_normalize=normalize)
        super().__init__(manifest_filepath, parser, sample_rate, int_values, augmentor, max_duration, min_duration, max_utts,
                         trim, bos_id, eos_id, pad_id, return_sample_id, channel_selector, manifest_parse_func)


class AudioToBPEDataset(_AudioTextDataset):
    def __init__(self, manifest_filepath: str, tokenizer, sample_rate: int, int_values: bool = False, augmentor=None,
                 max_duration: Optional[float] = None, min_duration: Optional[float] = None, max_utts: int = 0,
                 trim: bool = False, use_start_end_token: bool = True, return_sample_id: bool = False, channel_selector=None,
                 manifest_parse_func: Optional[Callable] = None):
        bos_id = tokenizer.bos_id if use_start_end_token and getattr(tokenizer, "bos_id", 0) > 0 else None
        eos_id = tokenizer.eos_id if use_start_end_token and getattr(tokenizer, "eos_id", 0) > 0 else None
        pad_id = tokenizer.pad_id if getattr(tokenizer, "pad_id", 0) > 0 else 0
        self.tokenizer = tokenizer
        super().__init__(manifest_filepath, TokenizerWrapper(tokenizer), sample_rate, int_values, augmentor, max_duration,
                         min_duration, max_utts, trim, bos_id, eos_id, pad_id, return_sample_id, channel_selector,
                         manifest_parse_func)
