"""Host-side sequence record: the fields the metadata producers read (nanovllm/engine/sequence.py:14-69)."""
from itertools import count


class Sequence:
    block_size = 256
    counter = count()

    def __init__(self, token_ids, max_tokens=64, ignore_eos=True, temperature=0.0):
        self.seq_id = next(Sequence.counter)
        self.token_ids = list(token_ids)
        self.last_token = self.token_ids[-1]
        self.num_tokens = len(self.token_ids)
        self.num_prompt_tokens = len(self.token_ids)
        self.num_cached_tokens = 0
        self.block_table = []
        self.max_tokens = max_tokens
        self.ignore_eos = ignore_eos
        self.temperature = temperature

    def __len__(self):
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    @property
    def num_completion_tokens(self):
        return self.num_tokens - self.num_prompt_tokens

    @property
    def completion_token_ids(self):
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_cached_blocks(self):
        return self.num_cached_tokens // self.block_size

    @property
    def num_blocks(self):
        return (self.num_tokens + self.block_size - 1) // self.block_size

    @property
    def last_block_num_tokens(self):
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def append_token(self, token_id):
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1
