"""Bags-of-items data model (mirrors the parts of reference aaerec/datasets.py that feed the AAE
path: helpers 19-123, Bags 126-414, BagsWithVocab 417-517).  A bag = the list of item tokens
owned by one document; attributes (side information) hang off the owner ids."""
import itertools as it
import random
from collections import Counter, defaultdict

import numpy as np

from .transforms import lists2sparse


def split_by_mask(data, condition):
    yes = [d for d, c in zip(data, condition) if c]
    no = [d for d, c in zip(data, condition) if not c]
    return yes, no


def build_vocab(sets, min_count=None, max_features=None):
    """token -> index by descending frequency; optional frequency floor / size cap."""
    counts = Counter(it.chain.from_iterable(sets)).most_common(max_features)
    if min_count:
        counts = list(it.takewhile(lambda kv: kv[1] >= min_count, counts))
    vocab = {}
    for token, _ in counts:
        vocab[token] = len(vocab)
    return vocab, counts


def filter_vocab(lists, vocab):
    return [[t for t in tokens if t in vocab] for tokens in lists]


def apply_vocab(lists, vocab):
    return [[vocab[t] for t in tokens] for tokens in lists]


def filter_apply_vocab(lists, vocab):
    return [[vocab[t] for t in tokens if t in vocab] for tokens in lists]


def filter_length(lists, min_length, *supplements):
    keep = [len(bag) >= min_length for bag in lists]
    out = [[x for x, k in zip(seq, keep) if k] for seq in (lists,) + supplements]
    return out[0] if not supplements else tuple(out)


def split_set(s, criterion):
    """(remaining, dropped): int -> drop that many random elements, float in (0,1) -> coin per
    element, callable -> drop where it returns True."""
    s = set(s)
    if callable(criterion):
        todrop = {e for e in s if criterion(e)}
    elif type(criterion) == float:
        assert 0 < criterion < 1, "Float not bounded in (0,1)"
        todrop = {e for e in s if random.random() < criterion}
    elif type(criterion) == int:
        try:
            todrop = random.sample(sorted(s), criterion)
        except ValueError:
            todrop = s
    else:
        raise ValueError("int, float, or callable expected")
    todrop = set(todrop)
    return s - todrop, todrop


def corrupt_sets(sets, drop=1):
    """([kept...], [dropped...]) for every set."""
    return tuple(zip(*[split_set(s, drop) for s in sets]))


class Bags:
    def __init__(self, data, owners, owner_attributes=None):
        assert len(owners) == len(data)
        self.data = data
        self.bag_owners = owners
        self.owner_attributes = owner_attributes

    def clone(self):
        attrs = None
        if self.owner_attributes is not None:
            attrs = {a: dict(by_owner) for a, by_owner in self.owner_attributes.items()}
        return Bags([list(b) for b in self.data], list(self.bag_owners), owner_attributes=attrs)

    def __len__(self):
        return len(self.data)

    def __str__(self):
        return "{} records with {} ratings".format(len(self), self.numel())

    def __getitem__(self, idx):
        return self.data[idx]

    def maxlen(self):
        return max(map(len, self.data))

    def numel(self):
        return sum(map(len, self.data))

    def get_single_attribute(self, attribute):
        if self.owner_attributes is None or self.bag_owners is None:
            raise ValueError("Owners not present")
        table = self.owner_attributes[attribute]
        return [table[o] for o in self.bag_owners]

    def get_attributes(self, attribute_list):
        return [self.get_single_attribute(a) for a in attribute_list]

    def to_dict(self):
        return dict(enumerate(self.data))

    @classmethod
    def load_tabcomma_format(cls, path, meta_data_dic=False, unique=False, owner_str="owner", set_str="set"):
        """TSV with a header row: one owner column, one comma-separated item-set column, every other
        column becomes an owner attribute (reference datasets.py:233-323, re-stated on the csv
        module so it does not depend on removed pandas keywords)."""
        import csv
        owners, sets = [], []
        attributes = defaultdict(dict)
        with open(path, newline="", encoding="utf-8") as fh:
            reader = csv.reader(fh, delimiter="\t", quoting=csv.QUOTE_NONE)
            header = next(reader)
            oi, si = header.index(owner_str), header.index(set_str)
            for row in reader:
                if len(row) != len(header):
                    continue
                owner = row[oi]
                tokens = [t for t in row[si].split(",") if t] if row[si] else []
                if unique:
                    tokens = list(dict.fromkeys(tokens))
                owners.append(owner)
                sets.append(tokens)
                for j, name in enumerate(header):
                    if j not in (oi, si):
                        v = row[j]
                        if name == "year":
                            try:
                                v = int(float(v))
                            except ValueError:
                                pass
                        attributes[name][owner] = v
        return cls(sets, owners, owner_attributes=dict(attributes))

    def train_test_split(self, on_year=None, **split_params):
        """on_year: owners whose 'year' attribute is < on_year train, the rest test."""
        if on_year is not None:
            assert self.owner_attributes["year"], "Cant split on non-existing 'year'"
            years = self.owner_attributes["year"]
            is_train = [int(years[o]) < on_year for o in self.bag_owners]
            tr_d, te_d = split_by_mask(self.data, is_train)
            tr_o, te_o = split_by_mask(self.bag_owners, is_train)
        else:
            from sklearn.model_selection import train_test_split
            tr_d, te_d, tr_o, te_o = train_test_split(self.data, self.bag_owners, **split_params)
        return (Bags(tr_d, tr_o, owner_attributes=self.owner_attributes),
                Bags(te_d, te_o, owner_attributes=self.owner_attributes))

    def build_vocab(self, min_count=None, max_features=None, apply=True):
        vocab, _ = build_vocab(self.data, min_count=min_count, max_features=max_features)
        return self.apply_vocab(vocab) if apply else vocab

    def apply_vocab(self, vocab):
        return BagsWithVocab(filter_apply_vocab(self.data, vocab), vocab, owners=self.bag_owners,
                             attributes=self.owner_attributes)

    def prune_(self, min_elements=0):
        if min_elements:
            if self.bag_owners is not None:
                self.data, self.bag_owners = filter_length(self.data, min_elements, self.bag_owners)
            else:
                self.data = filter_length(self.data, min_elements)
        return self

    def inflate(self, factor):
        self.data = [bag * factor for bag in self.data]
        return self


class BagsWithVocab(Bags):
    def __init__(self, data, vocab, owners=None, attributes=None):
        super().__init__(data, owners, owner_attributes=attributes)
        self.vocab = vocab
        self.index2token = {v: k for k, v in vocab.items()}

    def clone(self):
        attrs = None
        if self.owner_attributes is not None:
            attrs = {a: dict(by_owner) for a, by_owner in self.owner_attributes.items()}
        return BagsWithVocab([list(b) for b in self.data], dict(self.vocab), owners=list(self.bag_owners),
                             attributes=attrs)

    def build_vocab(self, min_count=None, max_features=None, apply=True):
        raise ValueError("Instance already has vocabulary.")

    def apply_vocab(self, vocab):
        raise ValueError("A vocabulary has already been applied.")

    def __str__(self):
        return "{} elements in [{}, {}] [data_points,vocabulary_size] with density {}".format(
            self.numel(), *self.size(), self.density())

    def size(self, dim=None):
        sizes = (len(self.data), len(self.vocab))
        return sizes if dim is None else sizes[dim]

    def tocsr(self, data=None):
        """scipy CSR float64 [n_docs, n_items]; values = multiplicity (reference 459-470)."""
        if data is None:
            data, size = self.data, self.size()
        else:
            size = (len(data), self.size(1))
        return lists2sparse(data, size).tocsr()

    def train_test_split(self, **split_params):
        tr, te = super().train_test_split(**split_params)
        return (BagsWithVocab(tr.data, self.vocab, owners=tr.bag_owners, attributes=tr.owner_attributes),
                BagsWithVocab(te.data, self.vocab, owners=te.bag_owners, attributes=te.owner_attributes))

    def density(self):
        return self.numel() / float(np.prod(self.size()))

    def raw(self):
        return apply_vocab(self.data, self.index2token)
