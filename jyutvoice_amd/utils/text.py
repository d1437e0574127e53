"""The token contract between the reference's text front-end and `synthesise()` -- the host half of SURVEY.md 8(f-3).

`infer.py:189-206` (`get_text`) turns `text_to_sequence(text, lang, phone)` (`jyutvoice/text/__init__.py:20-35`: five
equal-length lists -- phone ids, tones, word positions, syllable positions, language ids) into the six tensors
`synthesise()` takes by putting a blank (id 0) around every entry of each list (`intersperse`,
`jyutvoice/utils/utils.py:131-135`).  The G2P stack that produces the lists (pycantonese / pypinyin / g2p_en) is a CPU
string front-end outside this build; what IS the path's input contract lives here:

  intersperse(lst, item)        the reference's helper, same result
  validate_ids(...)             the five lists: equal lengths, ids inside the embedding tables of configs/base.yaml:65-67
                                (phone < 97, tone < 7, lang < 4) and text_encoder.py:372-375 (word_pos < 4, syllable_pos < 4);
                                with interspersed=True also the blank at every even position
  get_text_from_ids(...)        get_text() from the point where the G2P lists exist: intersperse + tensors, in get_text's
                                return order (x, x_lengths, tones, word_pos, syllable_pos, lang_ids)
  load_tokens_json(dict)        the --tokens file of infer.py here -> validated tensors (either list form)

An id outside its table is an out-of-bounds row of an embedding matrix in HBM: the kernels index without checks, so this is
where it is caught."""
from __future__ import annotations

from typing import Dict, Sequence

import torch

from .. import spec

# field -> (exclusive upper bound of the ids, where the table size comes from)
ID_RANGES = {
    "x": (spec.ENC_N_VOCAB, "n_vocab, configs/base.yaml:65"),
    "lang": (spec.ENC_N_LANG, "n_lang, configs/base.yaml:66"),
    "tone": (spec.ENC_N_TONE, "n_tone, configs/base.yaml:67"),
    "word_pos": (spec.ENC_N_WORD_POS, "text_encoder.py:374"),
    "syllable_pos": (spec.ENC_N_SYL_POS, "text_encoder.py:375"),
}
FIELDS = tuple(ID_RANGES)


def intersperse(lst, item):
    """[a, b] -> [item, a, item, b, item] (jyutvoice/utils/utils.py:131-135)"""
    result = [item] * (len(lst) * 2 + 1)
    result[1::2] = lst
    return result


def validate_ids(ids: Dict[str, Sequence[int]], interspersed: bool = True) -> int:
    """Raises ValueError naming the first violation; returns the common length."""
    missing = [k for k in FIELDS if k not in ids]
    if missing:
        raise ValueError(f"tokens: missing id lists {missing} (need {list(FIELDS)})")
    n = len(ids["x"])
    for k in FIELDS:
        v = ids[k]
        if len(v) != n:
            raise ValueError(f"tokens: '{k}' has {len(v)} entries, 'x' has {n}: the five lists must have equal length")
    if n == 0:
        raise ValueError("tokens: empty utterance")
    if interspersed and n % 2 == 0:
        raise ValueError(f"tokens: {n} entries, but an interspersed sequence has odd length (2 n + 1)")
    for k in FIELDS:
        hi, where = ID_RANGES[k]
        for i, t in enumerate(ids[k]):
            if isinstance(t, bool) or int(t) != t:
                raise ValueError(f"tokens: '{k}'[{i}] = {t!r} is not an integer id")
            if not 0 <= int(t) < hi:
                raise ValueError(f"tokens: '{k}'[{i}] = {t} is outside [0, {hi}) ({where})")
            if interspersed and i % 2 == 0 and int(t) != 0:
                raise ValueError(f"tokens: '{k}'[{i}] = {t}, but even positions hold the blank (id 0) after intersperse "
                                 f"(infer.py:194-198); pass interspersed=False for the raw text_to_sequence lists")
    return n


def get_text_from_ids(phone_token_ids, tones, word_pos, syllable_pos, lang_ids):
    """infer.py:189-206 from its second line on: the five text_to_sequence lists -> (x, x_lengths, tones, word_pos,
    syllable_pos, lang_ids), each list interspersed with the blank, as [1, 2 n + 1] int64 tensors."""
    raw = {"x": phone_token_ids, "tone": tones, "word_pos": word_pos, "syllable_pos": syllable_pos, "lang": lang_ids}
    validate_ids(raw, interspersed=False)
    phone_token_ids = intersperse(list(phone_token_ids), 0)
    tones = intersperse(list(tones), 0)
    word_pos = intersperse(list(word_pos), 0)
    syllable_pos = intersperse(list(syllable_pos), 0)
    lang_ids = intersperse(list(lang_ids), 0)
    x = torch.tensor([phone_token_ids])
    x_lengths = torch.tensor([len(phone_token_ids)])
    return x, x_lengths, torch.tensor([tones]), torch.tensor([word_pos]), torch.tensor([syllable_pos]), torch.tensor([lang_ids])


def load_tokens_json(tok: dict) -> Dict[str, torch.Tensor]:
    """The --tokens file: {"x", "lang", "tone", "word_pos", "syllable_pos"} as the INTERSPERSED lists (what get_text hands to
    synthesise), or with "interspersed": false as the raw text_to_sequence lists, which are interspersed here.
    -> {"x", "x_lengths", "lang", "tone", "word_pos", "syllable_pos"} int64 tensors [1, n] / [1]."""
    if bool(tok.get("interspersed", True)):
        validate_ids(tok, interspersed=True)
        out = {k: torch.tensor([list(map(int, tok[k]))], dtype=torch.int64) for k in FIELDS}
        out["x_lengths"] = torch.tensor([len(tok["x"])], dtype=torch.int64)
        return out
    x, x_lengths, tones, word_pos, syllable_pos, lang_ids = get_text_from_ids(tok["x"], tok["tone"], tok["word_pos"],
                                                                               tok["syllable_pos"], tok["lang"])
    return {"x": x, "x_lengths": x_lengths, "lang": lang_ids, "tone": tones, "word_pos": word_pos, "syllable_pos": syllable_pos}
