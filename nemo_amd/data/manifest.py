"""JSON-lines manifests -> the filtered, tokenised utterance list the datasets index.

Mirrors, for the ASR training path only:
  * `nemo/collections/common/parts/preprocessing/manifest.py:44-200` (`item_iter`, `__parse_item`, `get_full_path`): one JSON
    object per line; `audio_filepath` | `audio_filename`, `duration`, `text` | `text_filepath` | `normalized_text`,
    optional `offset`, `speaker`, `orig_sample_rate`, `token_labels`, `lang`; a relative audio path that does not exist
    is resolved against the manifest's directory; a line that is not JSON is an error reported after the whole file
    was read;
  * `collections.py:93-215,333-420` (`AudioText`, `ASRAudioText`): min / max duration filters, tokenisation through the
    parser (an utterance whose parser result is None is dropped), `max_utts` cap, the summary counters.
"""
from __future__ import annotations

import collections
import json
import os
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence, Union

AudioTextEntity = collections.namedtuple(
    "AudioTextEntity", "id audio_file duration text_tokens offset text_raw speaker orig_sr lang")


def get_full_path(audio_file: str, manifest_file: Optional[str] = None, audio_file_len_limit: int = 255) -> str:
    """manifest.py:203-300: an absolute path is kept; a relative one is tried against the manifest's directory and kept
    as written if that file does not exist either"""
    if len(audio_file) < audio_file_len_limit and not os.path.isabs(audio_file):
        if manifest_file is None:
            raise ValueError(f"Use of a relative path ({audio_file}) is not supported without a manifest file.")
        candidate = os.path.abspath(os.path.join(os.path.dirname(manifest_file), audio_file))
        if os.path.isfile(candidate):
            return candidate
    return audio_file


def parse_item(line: str, manifest_file: str) -> Dict[str, Any]:
    item = json.loads(line)
    if "audio_filename" in item:
        item["audio_file"] = item.pop("audio_filename")
    elif "audio_filepath" in item:
        item["audio_file"] = item.pop("audio_filepath")
    if "audio_file" not in item:
        raise ValueError(f"Manifest file {manifest_file} has invalid json line structure: {line} without proper audio file key.")
    item["audio_file"] = get_full_path(item["audio_file"], manifest_file)
    if "duration" not in item:
        raise ValueError(f"Manifest file {manifest_file} has invalid json line structure: {line} without proper duration key.")
    if "text" in item:
        pass
    elif "text_filepath" in item:
        with open(item.pop("text_filepath"), "r") as f:
            item["text"] = f.read().replace("\n", "")
    elif "normalized_text" in item:
        item["text"] = item["normalized_text"]
    else:
        item["text"] = ""
    return dict(audio_file=item["audio_file"], duration=item["duration"], text=item["text"], offset=item.get("offset"),
                speaker=item.get("speaker"), orig_sr=item.get("orig_sample_rate"), token_labels=item.get("token_labels"),
                lang=item.get("lang"))


def item_iter(manifests_files: Union[str, Sequence[str]], parse_func: Optional[Callable] = None) -> Iterator[Dict[str, Any]]:
    if isinstance(manifests_files, str):
        manifests_files = [manifests_files]
    parse_func = parse_func or parse_item
    errors: Dict[str, List[str]] = {}
    k = -1
    for manifest_file in manifests_files:
        with open(os.path.expanduser(manifest_file), "r") as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                k += 1
                try:
                    item = parse_func(line, manifest_file)
                except json.JSONDecodeError:
                    errors.setdefault(str(manifest_file), []).append(line)
                    continue
                item["id"] = k
                yield item
    if errors:
        detail = "; ".join(f"{len(v)} line(s) of {k_}" for k_, v in errors.items())
        raise RuntimeError(f"Failed to parse some lines from manifest files: {detail}")


class ASRAudioText:
    """the utterance list: `collection[i]` is an `AudioTextEntity`"""

    def __init__(self, manifests_files: Union[str, Sequence[str]], parser: Callable, min_duration: Optional[float] = None,
                 max_duration: Optional[float] = None, max_number: Optional[int] = None, do_sort_by_duration: bool = False,
                 parse_func: Optional[Callable] = None):
        if isinstance(manifests_files, str):
            manifests_files = manifests_files.split(",")  # "Can be comma-separated paths" (audio_to_text.py:401)
        self.data: List[AudioTextEntity] = []
        self.num_filtered, self.duration_filtered, self.total_duration = 0, 0.0, 0.0
        for item in item_iter(manifests_files, parse_func):
            duration = item["duration"]
            if duration is not None and min_duration is not None and duration < min_duration:
                self.duration_filtered += duration; self.num_filtered += 1
                continue
            if duration is not None and max_duration is not None and duration > max_duration:
                self.duration_filtered += duration; self.num_filtered += 1
                continue
            if item["token_labels"] is not None:
                tokens = item["token_labels"]
            else:
                text = item["text"]
                if text != "":
                    if getattr(parser, "is_aggregate", False) and isinstance(text, str):
                        if item["lang"] is None:
                            raise ValueError("lang required in manifest when using aggregate tokenizers")
                        tokens = parser(text, item["lang"])
                    else:
                        tokens = parser(text)
                else:
                    tokens = []
                if tokens is None:
                    self.duration_filtered += duration; self.num_filtered += 1
                    continue
            self.total_duration += duration if duration is not None else 0.0
            self.data.append(AudioTextEntity(item["id"], item["audio_file"], duration, tokens, item["offset"], item["text"],
                                             item["speaker"], item["orig_sr"], item["lang"]))
            if len(self.data) == max_number:
                break
        if do_sort_by_duration:
            self.data.sort(key=lambda e: e.duration)

    def __getitem__(self, i):
        return self.data[i]

    def __len__(self):
        return len(self.data)

    @property
    def durations(self) -> List[float]:
        return [e.duration for e in self.data]
