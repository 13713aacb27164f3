"""Physical KV block allocator: the allocation rules of nanovllm/engine/block_manager.py that define block-table
and slot semantics (allocate :62-115 without the prefix-cache hashing, may_append :134-159, deallocate :117-124).
Prefix caching (xxhash chains, ref counts) is host bookkeeping outside the attention path and is not rebuilt
here; paged prefill over cached prefixes is still exercised directly in the kernel tests."""
from collections import deque


class BlockManager:
    def __init__(self, num_blocks, block_size=256):
        self.block_size = block_size
        self.free_block_ids = deque(range(num_blocks))
        self.used_block_ids = set()

    def _take(self):
        block_id = self.free_block_ids.popleft()
        self.used_block_ids.add(block_id)
        return block_id

    def can_allocate(self, seq):
        return len(self.free_block_ids) >= seq.num_blocks

    def allocate(self, seq, reserve_tokens=0):
        """One physical block per logical block of the sequence; `reserve_tokens` pre-books room for tokens that
        will be generated (used by the graph-replayed decode session, whose block tables must be static)."""
        assert not seq.block_table
        need = (len(seq) + reserve_tokens + self.block_size - 1) // self.block_size
        if len(self.free_block_ids) < need:
            raise RuntimeError("out of memory: KV cache blocks exhausted")
        for _ in range(need):
            seq.block_table.append(self._take())

    def can_append(self, seq):
        return len(self.free_block_ids) >= (len(seq) % self.block_size == 1)

    def may_append(self, seq):
        """Called after a token was appended: a sequence that just spilled into a new block gets one."""
        if len(seq) % self.block_size == 1 and seq.num_blocks > len(seq.block_table):
            seq.block_table.append(self._take())

    def deallocate(self, seq):
        for block_id in reversed(seq.block_table):
            self.used_block_ids.discard(block_id)
            self.free_block_ids.append(block_id)
        seq.num_cached_tokens = 0
        seq.block_table.clear()
