"""Text front-end (tortoise_tts_amd/tokenizer.py) against the reference's VoiceBpeTokenizer run in the build container on its own vocabulary
(tests/golden/tokenizer.npz, tokenizer.py:154-177), and against the `tokenizers` library the reference delegates the BPE to."""
import json

import numpy as np
import pytest

from tortoise_tts_amd.tokenizer import VoiceBpeTokenizer, english_cleaners, number_to_words, ordinal_words, normalize_numbers, convert_to_ascii


def make(g):
	vocab = {str(t): i for i, t in enumerate(g["vocab"])}
	return VoiceBpeTokenizer(vocab=vocab, merges=[str(m) for m in g["merges"]], special_tokens=[str(s) for s in g["special"]])


def test_ids_equal_reference(golden):
	g = golden("tokenizer")
	tok = make(g)
	assert len(tok.get_vocab()) == 255 and tok.get_vocab()["[STOP]"] == 0 and tok.get_vocab()["[SPACE]"] == 2
	off = g["offsets"]
	assert len(g["texts"]) >= 50
	for k, text in enumerate(g["texts"]):
		want = g["ids"][off[k]:off[k + 1]].tolist()
		assert tok.preprocess_text(str(text)) == str(g["cleaned"][k]), repr(text)
		assert tok.encode(str(text)) == want, repr(text)
		assert tok.decode(np.array(want, dtype=np.int64)) == str(g["decoded"][k]), repr(text)
	assert tok.encode("") == []
	assert tok.encode("[STOP]") == tok.encode("[stop]") and 0 not in tok.encode("[STOP]")     # the cleaners lowercase first: not the special token


def test_file_loader_and_tokenizers_library(golden, tmp_path):
	"""the JSON loader (both merge spellings) and id equality with tokenizers.Tokenizer -- the third-party BPE the reference calls -- on random
	strings over the vocabulary's alphabet plus characters outside it"""
	tokenizers = pytest.importorskip("tokenizers")
	g = golden("tokenizer")
	vocab = {str(t): i for i, t in enumerate(g["vocab"])}
	merges = [str(m) for m in g["merges"]]
	spec = {"version": "1.0", "truncation": None, "padding": None, "normalizer": None, "pre_tokenizer": {"type": "Whitespace"}, "post_processor": None,
			"decoder": None,
			"added_tokens": [{"id": vocab[str(s)], "special": True, "content": str(s), "single_word": False, "lstrip": False, "rstrip": False, "normalized": False}
							 for s in g["special"]],
			"model": {"type": "BPE", "dropout": None, "unk_token": "[UNK]", "continuing_subword_prefix": None, "end_of_word_suffix": None, "fuse_unk": False,
					  "vocab": vocab, "merges": merges}}
	path = tmp_path / "tokenizer.json"
	path.write_text(json.dumps(spec))
	mine = VoiceBpeTokenizer(str(path))
	spec["model"]["merges"] = [m.split(" ") for m in merges]
	(tmp_path / "pairs.json").write_text(json.dumps(spec))
	mine_pairs = VoiceBpeTokenizer(str(tmp_path / "pairs.json"))
	lib = tokenizers.Tokenizer.from_file(str(path))
	rng = np.random.default_rng(5)
	alphabet = list("abcdefghijklmnopqrstuvwxyz") * 3 + list(" ,.'-!?;:()/_") + list("@#0159") + ["[SPACE]", "[STOP]", "th", "ing", "ou"]
	for n in rng.integers(1, 120, size=300):
		s = "".join(rng.choice(alphabet, size=int(n)))
		want = lib.encode(s.replace(" ", "[SPACE]")).ids
		raw = mine._special_re
		got = []
		pos = 0
		t = s.replace(" ", "[SPACE]")
		for m in raw.finditer(t):
			got += mine._encode_plain(t[pos:m.start()]) + [mine.vocab[m.group(0)]]
			pos = m.end()
		got += mine._encode_plain(t[pos:])
		assert got == want, repr(s)
	assert mine.encode("the thing") == mine_pairs.encode("the thing") == make(g).encode("the thing")
	with pytest.raises(ValueError):
		bad = dict(spec, pre_tokenizer={"type": "ByteLevel"})
		(tmp_path / "bad.json").write_text(json.dumps(bad))
		VoiceBpeTokenizer(str(tmp_path / "bad.json"))


def test_number_expansion_known_answers():
	"""inflect's published behaviour for the call forms of tokenizer.py:85-102 (the library is absent here: restated, not executed)"""
	assert number_to_words(0) == "zero" and number_to_words(7) == "seven" and number_to_words(13) == "thirteen" and number_to_words(40) == "forty"
	assert number_to_words(21) == "twenty-one" and number_to_words(100) == "one hundred"
	assert number_to_words(123) == "one hundred and twenty-three" and number_to_words(123, andword="") == "one hundred twenty-three"
	assert number_to_words(1234) == "one thousand, two hundred and thirty-four"
	assert number_to_words(1000000, andword="") == "one million" and number_to_words(45017, andword="") == "forty-five thousand seventeen"
	assert number_to_words(1001) == "one thousand and one" and number_to_words(45300, andword="") == "forty-five thousand, three hundred"
	# the Tacotron cleaners' own known answers for this pipeline (the reference's tokenizer.py is that code)
	for text, want in (("1", "one"), ("15", "fifteen"), ("24", "twenty-four"), ("100", "one hundred"), ("101", "one hundred one"), ("456", "four hundred fifty-six"),
					   ("1000", "one thousand"), ("1800", "eighteen hundred"), ("2,000", "two thousand"), ("3000", "three thousand"), ("18000", "eighteen thousand"),
					   ("24,000", "twenty-four thousand"), ("124,001", "one hundred twenty-four thousand one"), ("6.4 sec", "six point four sec"),
					   ("1906", "nineteen oh six"), ("2007", "two thousand seven"), ("1900", "nineteen hundred"), ("2010", "twenty ten"),
					   ("$3.50 for gas.", "three dollars, fifty cents for gas."), ("$1", "one dollar"), ("$20", "twenty dollars"), ("1st", "first"),
					   ("2nd", "second"), ("23rd", "twenty-third"), ("100th", "one hundredth")):
		assert normalize_numbers(text) == want, text
	assert number_to_words(1984, andword="", zero="oh", group=2) == "nineteen, eighty-four"
	assert number_to_words(1905, andword="", zero="oh", group=2) == "nineteen, oh five"
	assert ordinal_words(1) == "first" and ordinal_words(2) == "second" and ordinal_words(3) == "third" and ordinal_words(12) == "twelfth"
	assert ordinal_words(20) == "twentieth" and ordinal_words(21) == "twenty-first" and ordinal_words(100) == "one hundredth" and ordinal_words(45) == "forty-fifth"
	assert normalize_numbers("in 1984 and 2005, 1900 or 2000; 3000") == "in nineteen eighty-four and two thousand five, nineteen hundred or two thousand; three thousand"
	assert normalize_numbers("1,234 items") == "twelve thirty-four items"            # 1000 < n < 3000 reads as a year, tokenizer.py:91-100
	assert normalize_numbers("12,345 items") == "twelve thousand, three hundred forty-five items"
	assert normalize_numbers("$3.50 and $1 and $0.01") == "three dollars, fifty cents and one dollar and one cent"
	assert normalize_numbers("£20 for 2.5 kg on the 3rd") == "twenty pounds for two point five kg on the third"


def test_cleaners_and_transliteration():
	assert english_cleaners('Mr. Smith  said:\t"Hello"\nto Dr. No.') == "mister smith said: hello to doctor no."
	assert english_cleaners("St. Louis, Mo. Ltd. Co.") == "saint louis, mo. limited company"
	assert convert_to_ascii("plain") == "plain"
	assert convert_to_ascii("café naïve Ångström façade") == "cafe naive Angstrom facade"
	assert convert_to_ascii("“quoted” – it’s… straße") == '"quoted" - it\'s... strasse'
	assert english_cleaners("Æther & Œuvre") == "aether & oeuvre"
