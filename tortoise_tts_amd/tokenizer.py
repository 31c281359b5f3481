"""Text front-end of the hot path: `VoiceBpeTokenizer` of the reference (tokenizer.py:154-177) -- the
English cleaning pipeline (tokenizer.py:144-152) followed by a 255-entry BPE over a `tokenizers`-format vocabulary file.

Host logic, pure Python, no third-party imports: the reference leans on `tokenizers` (BPE), `inflect` (numbers to words) and `unidecode`
(ASCII transliteration); the last two are absent from this image, so their published behaviour is restated here for the call forms the
reference uses (tokenizer.py:85-102), and the BPE is restated so ids do not depend on a library version:
  * BPE: split off the added special tokens, pre-tokenise with the `Whitespace` rule (`\\w+|[^\\w\\s]+`), start from characters
    (unknown character -> `[UNK]`, never fused), then repeatedly merge the adjacent pair of lowest merge rank, leftmost first.
  * numbers: inflect's `number_to_words(n, andword=...)` grouping by thousands, `group=2` pair reading for years, ordinal suffixes.
  * transliteration: NFKD decomposition minus combining marks plus a table for the Latin letters and punctuation that do not decompose;
    anything else non-ASCII is dropped (unidecode would spell out other scripts: out of scope, and the vocabulary is English).
Parity: the BPE and the cleaning of digit-free ASCII text are pinned by tests/golden/tokenizer.npz (ids produced by the reference's own
class on its own vocabulary); number expansion and transliteration are restatements that could not be run against the absent libraries.
"""
from __future__ import annotations

import json
import re
import unicodedata
from typing import Dict, Iterable, List, Sequence, Tuple

_whitespace_re = re.compile(r"\s+")

_ABBREVIATIONS = [(re.compile("\\b%s\\." % a, re.IGNORECASE), b) for a, b in [
	("mrs", "misess"), ("mr", "mister"), ("dr", "doctor"), ("st", "saint"), ("co", "company"), ("jr", "junior"), ("maj", "major"),
	("gen", "general"), ("drs", "doctors"), ("rev", "reverend"), ("lt", "lieutenant"), ("hon", "honorable"), ("sgt", "sergeant"),
	("capt", "captain"), ("esq", "esquire"), ("ltd", "limited"), ("col", "colonel"), ("ft", "fort")]]    # tokenizer.py:17-36 (incl. 'misess')

# ------------------------------------------------------------------------------------------------ numbers to words (inflect's rules)
_UNIT = ["", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine"]
_TEEN = ["ten", "eleven", "twelve", "thirteen", "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen"]
_TEN = ["", "", "twenty", "thirty", "forty", "fifty", "sixty", "seventy", "eighty", "ninety"]
_MILL = ["", " thousand", " million", " billion", " trillion", " quadrillion", " quintillion", " sextillion", " septillion", " octillion",
		 " nonillion", " decillion"]
_ORDINAL_WORD = {"one": "first", "two": "second", "three": "third", "five": "fifth", "eight": "eighth", "nine": "ninth", "twelve": "twelfth"}


def _tens(t: int, u: int) -> str:
	if t == 1:
		return _TEEN[u]
	return _TEN[t] + ("-" if t and u else "") + _UNIT[u]


def number_to_words(num: int, andword: str = "and", zero: str = "zero", group: int = 0) -> str:
	"""inflect.engine().number_to_words for a non-negative integer.  group=0: thousands groups joined by ", ", `andword` between a
	group's hundreds and its remainder ("one thousand, two hundred and thirty-four") and in place of the last comma when the final group is
	one word.  group=2: the digits read in pairs, joined by
	", ", a leading zero of a pair spoken as `zero` ("nineteen, oh five")."""
	digits = str(int(num))
	if group == 2:
		parts = []
		for i in range(0, len(digits), 2):
			pair = digits[i:i + 2]
			if len(pair) == 1:
				parts.append(_UNIT[int(pair)] or zero)
			elif pair[0] == "0":
				parts.append(f"{zero} {_UNIT[int(pair[1])] or zero}")
			else:
				parts.append(_tens(int(pair[0]), int(pair[1])))
		return ", ".join(parts)
	if group != 0:
		raise NotImplementedError("only group=0 and group=2 are used by the cleaners")
	if int(digits) == 0:
		return zero
	chunks = []
	while digits:
		chunks.append(int(digits[-3:]))
		digits = digits[:-3]
	if len(chunks) > len(_MILL):
		raise ValueError("number out of range")
	words = []
	for mindex in range(len(chunks) - 1, -1, -1):
		h, rem = divmod(chunks[mindex], 100)
		t, u = divmod(rem, 10)
		if h:
			joint = (f" {andword} " if andword else " ") if rem else ""
			words.append(f"{_UNIT[h]} hundred{joint}{_tens(t, u)}{_MILL[mindex]}")
		elif rem:
			words.append(f"{_tens(t, u)}{_MILL[mindex]}")
	out = ", ".join(words)
	# inflect: a final group that is a single word ("one", "twenty-one") is joined with `andword` instead of the comma
	# ("one thousand and one"; with andword='' "one hundred twenty-four thousand one", the Tacotron cleaners' known answer for 124,001)
	head, sep, last = out.rpartition(", ")
	if sep and " " not in last:
		out = head + (f" {andword} " if andword else " ") + last
	return out


def ordinal_words(num: int) -> str:
	"""inflect's number_to_words("<n>th"): the cardinal (default `andword`) with its last word made ordinal."""
	card = number_to_words(num)
	head, sep, last = card.rpartition("-")
	if not sep:
		head, sep, last = card.rpartition(" ")
	if last in _ORDINAL_WORD:
		last = _ORDINAL_WORD[last]
	elif last.endswith("y"):
		last = last[:-1] + "ieth"
	else:
		last = last + "th"
	return head + sep + last


_comma_number_re = re.compile(r"([0-9][0-9\,]+[0-9])")
_decimal_number_re = re.compile(r"([0-9]+\.[0-9]+)")
_pounds_re = re.compile(r"£([0-9\,]*[0-9]+)")
_dollars_re = re.compile(r"\$([0-9\.\,]*[0-9]+)")
_ordinal_re = re.compile(r"[0-9]+(st|nd|rd|th)")
_number_re = re.compile(r"[0-9]+")


def _expand_dollars(m) -> str:   # tokenizer.py:61-82
	match = m.group(1)
	parts = match.split(".")
	if len(parts) > 2:
		return match + " dollars"
	dollars = int(parts[0]) if parts[0] else 0
	cents = int(parts[1]) if len(parts) > 1 and parts[1] else 0
	du, cu = ("dollar" if dollars == 1 else "dollars"), ("cent" if cents == 1 else "cents")
	if dollars and cents:
		return f"{dollars} {du}, {cents} {cu}"
	if dollars:
		return f"{dollars} {du}"
	if cents:
		return f"{cents} {cu}"
	return "zero dollars"


def _expand_number(m) -> str:    # tokenizer.py:89-102
	num = int(m.group(0))
	if 1000 < num < 3000:
		if num == 2000:
			return "two thousand"
		if 2000 < num < 2010:
			return "two thousand " + number_to_words(num % 100)
		if num % 100 == 0:
			return number_to_words(num // 100) + " hundred"
		return number_to_words(num, andword="", zero="oh", group=2).replace(", ", " ")
	return number_to_words(num, andword="")


def normalize_numbers(text: str) -> str:   # tokenizer.py:105-112
	text = re.sub(_comma_number_re, lambda m: m.group(1).replace(",", ""), text)
	text = re.sub(_pounds_re, r"\1 pounds", text)
	text = re.sub(_dollars_re, _expand_dollars, text)
	text = re.sub(_decimal_number_re, lambda m: m.group(1).replace(".", " point "), text)
	text = re.sub(_ordinal_re, lambda m: ordinal_words(int(re.match(r"[0-9]+", m.group(0)).group(0))), text)
	text = re.sub(_number_re, _expand_number, text)
	return text


# ------------------------------------------------------------------------------------------------ transliteration
_TRANSLIT = {
	"ß": "ss", "æ": "ae", "Æ": "AE", "œ": "oe", "Œ": "OE", "ø": "o", "Ø": "O", "đ": "d", "Đ": "D", "ð": "d", "Ð": "D", "þ": "th", "Þ": "Th",
	"ł": "l", "Ł": "L", "ı": "i", "ħ": "h", "Ħ": "H", "‘": "'", "’": "'", "‚": ",", "“": '"', "”": '"', "„": '"', "–": "-", "—": "--", "―": "--",
	"…": "...", "«": "<<", "»": ">>", "‹": "<", "›": ">", "·": "*", "•": "*", "×": "x", "÷": "/", "¡": "!", "¿": "?", "€": "EUR", "©": "(c)",
	"®": "(r)", "™": "(tm)", "°": "deg", "№": "No", "\u00a0": " ", "\u2002": " ", "\u2003": " ", "\u2009": " ", "\u200a": " ", "\u202f": " ",
}


def convert_to_ascii(text: str) -> str:
	if text.isascii():
		return text
	out = []
	for ch in text:
		if ord(ch) < 128:
			out.append(ch)
		elif ch in _TRANSLIT:
			out.append(_TRANSLIT[ch])
		else:
			for d in unicodedata.normalize("NFKD", ch):
				if ord(d) < 128:
					out.append(d)
				elif d in _TRANSLIT:
					out.append(_TRANSLIT[d])
	return "".join(out)


def english_cleaners(text: str) -> str:
	"""tokenizer.py:144-152."""
	text = convert_to_ascii(text)
	text = text.lower()
	text = normalize_numbers(text)
	for regex, replacement in _ABBREVIATIONS:
		text = re.sub(regex, replacement, text)
	text = re.sub(_whitespace_re, " ", text)
	return text.replace('"', "")


# ------------------------------------------------------------------------------------------------ BPE
_pretok_re = re.compile(r"\w+|[^\w\s]+")


class VoiceBpeTokenizer:
	"""`VoiceBpeTokenizer(tokenizer_file)` of the reference: `encode(text) -> List[int]`, `decode(ids) -> str`, `get_vocab()`."""

	def __init__(self, tokenizer_file: str = None, *, vocab: Dict[str, int] = None, merges: Sequence = None, special_tokens: Iterable[str] = (),
				 unk_token: str = "[UNK]"):
		if tokenizer_file is not None:
			with open(tokenizer_file, "r", encoding="utf-8") as f:
				j = json.load(f)
			model = j["model"]
			if model.get("type", "BPE") != "BPE" or (j.get("pre_tokenizer") or {}).get("type") != "Whitespace" or j.get("normalizer") is not None:
				raise ValueError("tokenizer file is not a plain Whitespace-pre-tokenised BPE (the TorToiSe vocabulary layout)")
			vocab, merges, unk_token = model["vocab"], model["merges"], model.get("unk_token") or unk_token
			special_tokens = [a["content"] for a in j.get("added_tokens", [])]
		if vocab is None or merges is None:
			raise ValueError("need a tokenizer file, or vocab= and merges=")
		self.vocab = dict(vocab)
		self.unk_token = unk_token
		self.ranks: Dict[Tuple[str, str], int] = {}
		for r, m in enumerate(merges):
			a, b = m.split(" ") if isinstance(m, str) else m      # "t h" (older files) or ["t", "h"]
			self.ranks.setdefault((a, b), r)
		self.special = [s for s in special_tokens if s in self.vocab]
		self._special_re = re.compile("|".join(re.escape(s) for s in sorted(self.special, key=len, reverse=True))) if self.special else None
		self.inv = {i: tok for tok, i in self.vocab.items()}

	def preprocess_text(self, txt: str) -> str:
		return english_cleaners(txt)

	def _bpe_word(self, word: str) -> List[int]:
		parts = list(word)
		while len(parts) > 1:
			best, at = None, -1
			for i in range(len(parts) - 1):
				r = self.ranks.get((parts[i], parts[i + 1]))
				if r is not None and (best is None or r < best):
					best, at = r, i
			if best is None:
				break
			parts[at:at + 2] = [parts[at] + parts[at + 1]]
		unk = self.vocab.get(self.unk_token)
		ids = []
		for p in parts:
			if p in self.vocab:
				ids.append(self.vocab[p])
			elif unk is not None:
				ids.append(unk)
		return ids

	def _encode_plain(self, text: str) -> List[int]:
		ids: List[int] = []
		for word in _pretok_re.findall(text):
			ids.extend(self._bpe_word(word))
		return ids

	def encode(self, txt: str) -> List[int]:
		"""tokenizer.py:163-166."""
		txt = self.preprocess_text(txt).replace(" ", "[SPACE]")
		if self._special_re is None:
			return self._encode_plain(txt)
		ids: List[int] = []
		pos = 0
		for m in self._special_re.finditer(txt):
			ids.extend(self._encode_plain(txt[pos:m.start()]))
			ids.append(self.vocab[m.group(0)])
			pos = m.end()
		ids.extend(self._encode_plain(txt[pos:]))
		return ids

	def decode(self, seq) -> str:
		"""tokenizer.py:168-174."""
		if hasattr(seq, "cpu"):
			seq = seq.cpu().numpy()
		txt = "".join(self.inv.get(int(i), "") for i in seq)      # tokens joined by ' ' and the spaces removed again
		return txt.replace("[SPACE]", " ").replace("[STOP]", "").replace("[UNK]", "")

	def get_vocab(self) -> Dict[str, int]:
		return dict(self.vocab)
