"""Transcript -> token ids.

`CharParser` mirrors `nemo/collections/common/parts/preprocessing/parsers.py:23-116` (label list = id map, lower-casing,
multi-character "special" labels matched per word, out-of-vocabulary -> `unk_id`, ids equal to `blank_id` removed).
`SentencePieceTokenizer` is the slice of `common/tokenizers/sentencepiece_tokenizer.py` that the BPE dataset and the
decoders use (`text_to_ids`, `ids_to_text`, `ids_to_tokens`, `vocab_size`, bos / eos / pad ids); `TokenizerWrapper` is
the adapter `AudioToBPEDataset` puts around it (audio_to_text.py:684-700).
"""
from __future__ import annotations

from typing import List, Optional


class CharParser:
    def __init__(self, labels: List[str], *, unk_id: int = -1, blank_id: int = -1, do_normalize: bool = True,
                 do_lowercase: bool = True, do_tokenize: bool = True):
        self._labels = labels
        self._unk_id, self._blank_id = unk_id, blank_id
        self._do_normalize, self._do_lowercase, self._do_tokenize = do_normalize, do_lowercase, do_tokenize
        self._labels_map = {label: index for index, label in enumerate(labels)}
        self._special_labels = {label for label in labels if len(label) > 1}

    def __call__(self, text: str) -> Optional[List[int]]:
        if self._do_normalize:
            text = text.strip()
            if self._do_lowercase:
                text = text.lower()
        if not self._do_tokenize:
            return text
        tokens: List[int] = []
        for word_id, word in enumerate(text.split(" ")):
            if word_id != 0:
                tokens.append(self._labels_map.get(" ", self._unk_id))
            if word in self._special_labels:
                tokens.append(self._labels_map[word])
                continue
            for char in word:
                tokens.append(self._labels_map.get(char, self._unk_id))
        return [t for t in tokens if t != self._blank_id]  # unk_id == blank_id removes the OOV symbols

    def decode(self, ids) -> str:
        r_map = {v: k for k, v in self._labels_map.items()}
        return "".join(r_map[int(i)] for i in ids if int(i) in r_map)


def make_parser(labels: List[str], name: str = "base", unk_id: int = -1, blank_id: int = -1, do_normalize: bool = True):
    """parsers.make_parser (parsers.py:270-310) for the language-independent 'base' parser; the English parser's number /
    abbreviation expansion needs `inflect` + `text_unidecode`, which the recipes only use for char models with
    `normalize_transcripts=True`."""
    if name not in ("base", None):
        raise NotImplementedError(f"parser '{name}': only the 'base' character parser is provided")
    return CharParser(labels, unk_id=unk_id, blank_id=blank_id, do_normalize=do_normalize)


class SentencePieceTokenizer:
    def __init__(self, model_path: str):
        import sentencepiece
        self.tokenizer = sentencepiece.SentencePieceProcessor()
        self.tokenizer.Load(model_path)
        self.vocab_size = self.tokenizer.get_piece_size()
        self.bos_id, self.eos_id = self.tokenizer.bos_id(), self.tokenizer.eos_id()
        self.pad_id, self.unk_id = self.tokenizer.pad_id(), self.tokenizer.unk_id()

    def text_to_ids(self, text: str) -> List[int]:
        return self.tokenizer.encode_as_ids(text)

    def text_to_tokens(self, text: str) -> List[str]:
        return self.tokenizer.encode_as_pieces(text)

    def ids_to_text(self, ids) -> str:
        return self.tokenizer.decode_ids([int(i) for i in ids])

    def ids_to_tokens(self, ids) -> List[str]:
        return [self.tokenizer.id_to_piece(int(i)) for i in ids]

    @property
    def vocab(self) -> List[str]:
        return [self.tokenizer.id_to_piece(i) for i in range(self.vocab_size)]


class TokenizerWrapper:
    def __init__(self, tokenizer):
        self.is_aggregate = False
        self._tokenizer = tokenizer

    def __call__(self, *args):
        return self._tokenizer.text_to_ids(*args)
