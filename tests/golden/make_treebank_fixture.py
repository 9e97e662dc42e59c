"""Generates tests/golden/treebank_tokenize.json.gz: (sentence, tokens) pairs from NLTK's NLTKWordTokenizer - the
per-sentence half of `word_tokenize` (keywords_search.py:13-18 upstream calls nltk.tokenize.word_tokenize).

Run with an interpreter that has nltk (this image: /opt/conda/bin/python3.9, nltk 3.6.5; the reference pins 3.9.1).
NLTK is a third-party dependency of the reference, not a reference file; the tokenizer needs no data files (Punkt, the
sentence splitter in front of it, does - and those are absent here, see aidial_rag_amd/keywords_search.py).
Sentences: every line of this repository's Markdown files and of a slice of the standard library's docstrings, plus
synthetic ones built around the constructs the rules test (quotes of every kind, clitics, contractions, commas and
colons before digits, ellipses, final periods behind brackets and quotes, brackets, double dashes, symbols).
"""
import glob, gzip, itertools, json, os, random, re

from nltk.tokenize.destructive import NLTKWordTokenizer
import nltk

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tok = NLTKWordTokenizer()
sents = set()
for f in sorted(glob.glob(os.path.join(root, "*.md"))) + sorted(glob.glob("/usr/lib/python3.10/*.py"))[:120]:
    try:
        text = open(f, encoding="utf-8", errors="ignore").read()
    except OSError:
        continue
    for line in text.split("\n"):
        line = line.strip()
        if 3 <= len(line) <= 300:
            sents.add(line)
sents = set(sorted(sents)[::3])
rnd = random.Random(7)
words = ["the", "Alps", "can't", "cannot", "gonna", "wanna go", "gimme", "lemme", "d'ye", "more'n", "'tis", "'twas", "it's", "I'm", "they'll", "we've",
         "don't", "DON'T", "you're", "he'd", "rock 'n' roll", "O'Neil", "'em", "'cause", "3,000", "1:30", "a,b", "x:y", "U.S.A.", "e.g.", "Mr. Smith",
         "etc.", "wait...", "no....", "..", "end.", "end.)", "end.\"", "end.'", "(paren)", "[bracket]", "{brace}", "<tag>", "a--b", "--", "50%", "$5", "#1",
         "a@b.c", "R&D", "semi;colon", "why?", "what!", "star*", "\"quoted\"", "''double''", "``back``", "`tick`", "«guillemet»", "“curly”", "‘single’",
         "„low“", "it's'", "s'", "dogs' bones", "the 'quote'", "\"Start", "say \"hi\"", "x,", "y:", "1,", "2:", ",lead", ":lead", "naïve café", "emoji 😀"]
for _ in range(6000):
    n = rnd.randint(1, 9)
    s = " ".join(rnd.choice(words) for _ in range(n))
    if rnd.random() < 0.3:
        s += rnd.choice([".", "!", "?", ".\"", ".)", "...", ":", ",", "'", "\"", " "])
    if rnd.random() < 0.1:
        s = rnd.choice(["\"", "'", "(", " ", "``"]) + s
    sents.add(s)
for a, b in itertools.product(["", " ", "x", "\"", "'", "(", ")"], repeat=2):
    sents.add(a + "word" + b)
    sents.add(a + "word." + b)
pairs = [[s, tok.tokenize(s)] for s in sorted(sents)]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "treebank_tokenize.json.gz")
with gzip.open(out, "wt", encoding="utf-8") as f:
    json.dump({"nltk_version": nltk.__version__, "pairs": pairs}, f, ensure_ascii=False)
print(len(pairs), "pairs ->", out)
