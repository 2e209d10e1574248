"""ORACLE (test infrastructure only): skeleton of a coalesced block matrix, densify, damp.

Pure-Python/numpy restatement of baspacho/baspacho/CoalescedBlockMatrix.cpp:
  constructor :17-122, densify :124-158, damp :172-187.
"""
import numpy as np

K_INVALID = -1


def build_skeleton(span_start, lump_to_span, col_ptr, row_ind):
    """CoalescedBlockMatrixSkel::CoalescedBlockMatrixSkel (CoalescedBlockMatrix.cpp:17-122)."""
    span_start = [int(x) for x in span_start]
    lump_to_span = [int(x) for x in lump_to_span]
    col_ptr = [int(x) for x in col_ptr]
    row_ind = [int(x) for x in row_ind]
    n_spans = len(span_start) - 1
    n_lumps = len(lump_to_span) - 1
    assert lump_to_span[-1] == n_spans and len(col_ptr) == len(lump_to_span)

    span_to_lump = [0] * (n_spans + 1)
    lump_start = [0] * (n_lumps + 1)
    for l in range(n_lumps):                               # :34-41
        lump_start[l] = span_start[lump_to_span[l]]
        for s in range(lump_to_span[l], lump_to_span[l + 1]):
            span_to_lump[s] = l
    span_to_lump[n_spans] = n_lumps                        # :42
    lump_start[n_lumps] = span_start[n_spans]              # :43
    span_offset_in_lump = [span_start[s] - lump_start[span_to_lump[s]] for s in range(n_spans)]
    span_offset_in_lump.append(0)                          # :44-48

    chain_col_ptr, chain_row_span, chain_data, chain_rows_till_end = [], [], [], []
    board_col_ptr, board_row_lump, board_chain_col_ord = [], [], []
    data_ptr = 0
    for l in range(n_lumps):                               # :59-100
        c_start, c_end = col_ptr[l], col_ptr[l + 1]
        l_span_begin, l_span_end = lump_to_span[l], lump_to_span[l + 1]
        l_data_size = lump_start[l + 1] - lump_start[l]
        assert c_end - c_start >= l_span_end - l_span_begin
        assert row_ind[c_start] == l_span_begin
        assert row_ind[c_start + (l_span_end - l_span_begin) - 1] == l_span_end - 1
        chain_col_ptr.append(len(chain_row_span))
        board_col_ptr.append(len(board_row_lump))
        current_row_aggreg = K_INVALID
        rows_skipped = 0
        for i in range(c_start, c_end):
            p = row_ind[i]
            chain_row_span.append(p)
            chain_data.append(data_ptr)
            data_ptr += l_data_size * (span_start[p + 1] - span_start[p])
            rows_skipped += span_start[p + 1] - span_start[p]
            chain_rows_till_end.append(rows_skipped)
            row_aggreg = span_to_lump[p]
            if row_aggreg != current_row_aggreg:
                current_row_aggreg = row_aggreg
                board_row_lump.append(row_aggreg)
                board_chain_col_ord.append(i - c_start)
        board_row_lump.append(K_INVALID)
        board_chain_col_ord.append(c_end - c_start)
    chain_col_ptr.append(len(chain_row_span))              # :101-103
    board_col_ptr.append(len(board_row_lump))
    chain_data.append(data_ptr)

    board_row_ptr = [0] * (n_lumps + 1)                    # :105-121
    for l in range(n_lumps):
        for i in range(board_col_ptr[l], board_col_ptr[l + 1] - 1):
            board_row_ptr[board_row_lump[i]] += 1
    tot = 0
    for l in range(n_lumps + 1):
        board_row_ptr[l], tot = tot, tot + (board_row_ptr[l] if l < n_lumps else 0)
    n_boards = board_row_ptr[n_lumps]
    board_col_lump = [0] * n_boards
    board_col_ord = [0] * n_boards
    cursor = list(board_row_ptr[:-1])
    for l in range(n_lumps):
        for i in range(board_col_ptr[l], board_col_ptr[l + 1] - 1):
            r = board_row_lump[i]
            board_col_lump[cursor[r]] = l
            board_col_ord[cursor[r]] = i - board_col_ptr[l]
            cursor[r] += 1

    as_arr = lambda v: np.asarray(v, dtype=np.int64)
    return {
        "spanStart": as_arr(span_start), "spanToLump": as_arr(span_to_lump),
        "lumpStart": as_arr(lump_start), "lumpToSpan": as_arr(lump_to_span),
        "spanOffsetInLump": as_arr(span_offset_in_lump),
        "chainColPtr": as_arr(chain_col_ptr), "chainRowSpan": as_arr(chain_row_span),
        "chainData": as_arr(chain_data), "chainRowsTillEnd": as_arr(chain_rows_till_end),
        "boardColPtr": as_arr(board_col_ptr), "boardRowLump": as_arr(board_row_lump),
        "boardChainColOrd": as_arr(board_chain_col_ord),
        "boardRowPtr": as_arr(board_row_ptr), "boardColLump": as_arr(board_col_lump),
        "boardColOrd": as_arr(board_col_ord),
    }


def order(sk):
    return int(sk["spanStart"][-1])


def data_size(sk):
    return int(sk["chainData"][-1])


def densify(sk, data, fill_upper_half=False, start_span_index=0):
    """CoalescedBlockMatrixSkel::densify (CoalescedBlockMatrix.cpp:124-158)."""
    ss, s2l, ls = sk["spanStart"], sk["spanToLump"], sk["lumpStart"]
    ccp, crs, cd = sk["chainColPtr"], sk["chainRowSpan"], sk["chainData"]
    assert sk["spanOffsetInLump"][start_span_index] == 0
    offset = int(ss[start_span_index])
    n = int(ss[-1]) - offset
    data = np.asarray(data)
    dense = np.zeros((n, n), dtype=data.dtype)
    for a in range(int(s2l[start_span_index]), len(ccp) - 1):
        l_begin = int(ls[a])
        l_size = int(ls[a + 1]) - l_begin
        for i in range(int(ccp[a]), int(ccp[a + 1])):
            p = int(crs[i])
            p_start = int(ss[p])
            p_size = int(ss[p + 1]) - p_start
            dp = int(cd[i])
            dense[p_start - offset:p_start - offset + p_size,
                  l_begin - offset:l_begin - offset + l_size] = \
                data[dp:dp + p_size * l_size].reshape(p_size, l_size)
    if fill_upper_half:
        low = np.tril(dense, -1)
        dense = np.tril(dense) + low.T
    return dense


def damp(sk, data, alpha, beta):
    """CoalescedBlockMatrixSkel::damp (CoalescedBlockMatrix.cpp:172-187); in place."""
    ls, ccp, cd = sk["lumpStart"], sk["chainColPtr"], sk["chainData"]
    assert len(data) == data_size(sk)
    for a in range(len(ccp) - 1):
        size = int(ls[a + 1] - ls[a])
        dp = int(cd[ccp[a]])
        idx = dp + np.arange(size) * (size + 1)
        data[idx] = data[idx] * (1 + alpha) + beta
    return data
