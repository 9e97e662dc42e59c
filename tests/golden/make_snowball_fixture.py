"""Generates tests/golden/snowball_english.json.gz: (token, stem) pairs from NLTK's SnowballStemmer("english").

Run with an interpreter that has nltk (this image: /opt/conda/bin/python3.9, nltk 3.6.5; the reference pins
3.9.1, pyproject.toml).  NLTK is a third-party dependency of the reference (keywords_search.py:1-11), not a
reference file.  Words: every alphabetic word of the Python standard library's sources and of this repository's
Markdown files, plus synthetic words built from stems x every suffix the algorithm knows, apostrophes, y/Y cases,
the gener/commun/arsen prefixes, non-ASCII letters, non-words and 20 000 random strings over the letters the rules test.
"""
import glob, gzip, itertools, json, os, re, sys

import nltk
from nltk.stem.snowball import SnowballStemmer

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
words = set()
files = sorted(glob.glob("/usr/lib/python3.10/*.py"))[:400] + sorted(glob.glob(os.path.join(root, "*.md")))
for f in files:
    try:
        text = open(f, encoding="utf-8", errors="ignore").read()
    except OSError:
        continue
    words.update(w.lower() for w in re.findall(r"[A-Za-z][A-Za-z'’]{0,24}", text))
words = set(sorted(words)[::2])  # every other one: ~20k is plenty

stems = ["", "a", "b", "tr", "hop", "agre", "relat", "gener", "commun", "arsen", "generat", "communic", "y", "say", "cry",
         "bay", "boy", "enjoy", "fl", "sk", "oper", "ration", "sens", "hope", "fil", "fizz", "add", "egg", "run", "plan",
         "conflat", "troubl", "siz", "luxuri", "necess", "bow", "box", "ax", "succe", "proce", "anal", "olog", "geolog",
         "trilog", "able", "feas", "respons", "use", "care", "hope", "form", "nation", "loc", "electr", "activ", "decis",
         "adopt", "adjust", "depend", "repl", "irrit", "differ", "effect", "bown", "real", "mate", "rate", "ceas", "live",
         "controll", "roll", "e", "ee", "fee", "lie", "tie", "die", "crie", "ü", "naïv", "café", "x1", "3", "--"]
sufs = ["", "s", "'s", "'s'", "'", "’s", "sses", "ied", "ies", "us", "ss", "eedly", "ingly", "edly", "eed", "ing", "ed",
        "ization", "ational", "fulness", "ousness", "iveness", "tional", "biliti", "lessli", "entli", "ation", "alism",
        "aliti", "ousli", "iviti", "fulli", "enci", "anci", "abli", "izer", "ator", "alli", "bli", "ogi", "li", "alize",
        "icate", "iciti", "ative", "ical", "ness", "ful", "ement", "ance", "ence", "able", "ible", "ment", "ant", "ent",
        "ism", "ate", "iti", "ous", "ive", "ize", "ion", "sion", "tion", "al", "er", "ic", "e", "l", "ll", "y", "ly",
        "ily", "ying", "yed", "ys", "ey", "ay", "ely", "fully", "lessly", "ization's", "ationally", "ousnesses"]
for a, b in itertools.product(stems, sufs):
    words.add(a + b)
    words.add("'" + a + b)
    words.add("y" + a + b)
for a, b, c in itertools.product(["gener", "commun", "arsen", "un", "re"], ["at", "ic", "al", "ous", "iv"], sufs[:60]):
    words.add(a + b + c)
words.update([",", ".", "``", "''", "n't", "'re", "'ll", "i.e.", "e.g.", "u.s.a", "co-operate", "state-of-the-art", "1990s",
              "3rd", "‘quoted’", "‛x", "ﬁnally", "ångström", "日本語", "日本語ing", "ied", "ies", "yyy", "ayayay"])
import random
rnd = random.Random(1)  # random short strings over the letters the rules look at: they reach the corners real words miss
alpha = "aeiouybcdlnrst'gmz"
for _ in range(20000):
    words.add("".join(rnd.choice(alpha) for _ in range(rnd.randint(0, 9))))
stemmer = SnowballStemmer("english")
pairs = [[w, stemmer.stem(w)] for w in sorted(words)]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "snowball_english.json.gz")
with gzip.GzipFile(out, "wb", mtime=0) as f:
    f.write(json.dumps({"source": f"nltk {nltk.__version__} SnowballStemmer('english').stem", "pairs": pairs},
                       ensure_ascii=False).encode("utf-8"))
print(len(pairs), "pairs ->", out, os.path.getsize(out), "bytes")
