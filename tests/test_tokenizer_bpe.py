"""The in-tree CLIP byte-pair encoder (lseg_hip.tokenizer.BPETokenizer, restating [3P] CLIP@04f4dc2
clip/simple_tokenizer.py) against an independent BPE implementation (HF `tokenizers`) on a synthetic merges file --
the real bpe_simple_vocab_16e6.txt.gz is not available offline, the algorithm is what is checked here: greedy
lowest-rank merges, the `</w>` end-of-word marker, lower-casing / whitespace collapsing, id layout
[256 bytes | 256 bytes+</w> | merges | SOT | EOT]."""
import gzip
import os

import pytest

from lseg_hip.tokenizer import BPETokenizer, _bytes_to_unicode

CORPUS = ("wall building sky floor tree ceiling road bed windowpane grass cabinet sidewalk person earth door table "
          "mountain plant curtain chair car water painting sofa shelf house sea mirror rug field armchair seat fence "
          "desk rock wardrobe lamp bathtub railing cushion base box column signboard chest counter sand sink "
          "skyscraper fireplace refrigerator grandstand path stairs runway case pool pillow screen stairway river "
          "bridge bookcase blind coffee toilet flower book hill bench countertop stove palm kitchen computer swivel "
          "boat bar arcade hovel bus towel light truck tower chandelier awning streetlight booth television airplane").split()


def _learn_merges(words, n):
    """Plain BPE training (most frequent adjacent pair first): gives a realistic rank table."""
    seqs = [tuple(w[:-1]) + (w[-1] + "</w>",) for w in words]
    merges = []
    for _ in range(n):
        cnt = {}
        for s in seqs:
            for a, b in zip(s, s[1:]):
                cnt[(a, b)] = cnt.get((a, b), 0) + 1
        if not cnt:
            break
        best = max(sorted(cnt), key=lambda p: cnt[p])
        merges.append(best)
        new = []
        for s in seqs:
            out, i = [], 0
            while i < len(s):
                if i < len(s) - 1 and (s[i], s[i + 1]) == best:
                    out.append(s[i] + s[i + 1]); i += 2
                else:
                    out.append(s[i]); i += 1
            new.append(tuple(out))
        seqs = new
    return merges


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    merges = _learn_merges(CORPUS, 120)
    path = os.path.join(tmp_path_factory.mktemp("bpe"), "vocab.txt.gz")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: synthetic\n" + "\n".join(" ".join(m) for m in merges) + "\n")
    return BPETokenizer(path), merges


def test_id_layout_and_specials(tok):
    t, merges = tok
    b2u = _bytes_to_unicode()
    assert t.enc[b2u[ord("a")]] < 256 and t.enc[b2u[ord("a")] + "</w>"] == t.enc[b2u[ord("a")]] + 256
    assert t.enc["".join(merges[0])] == 512
    assert t.enc["<|startoftext|>"] == 512 + len(t.ranks) and t.enc["<|endoftext|>"] == 513 + len(t.ranks)


def test_merges_match_an_independent_bpe(tok):
    tokenizers = pytest.importorskip("tokenizers")
    t, merges = tok
    vocab = dict(t.enc)
    model = tokenizers.models.BPE(vocab=vocab, merges=[tuple(m) for m in merges], end_of_word_suffix="</w>")
    hf = tokenizers.Tokenizer(model)
    for w in CORPUS + ["skywall", "treehouse", "zzz", "a", "windowpanes"]:
        ours = t.encode(w)
        theirs = hf.encode(w).ids
        assert ours == theirs, (w, ours, theirs)


def test_text_normalisation(tok):
    t, _ = tok
    assert t.encode("  Tree   HOUSE ") == t.encode("tree") + t.encode("house")     # lower-case, collapsed whitespace
    assert t.encode("tree&amp;house") == t.encode("tree") + t.encode("&") + t.encode("house")   # html.unescape
    assert t.encode("") == []
