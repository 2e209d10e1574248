/* ORACLE (test infrastructure only): skeleton arrays handed to the C restatement.
   Field names = CoalescedBlockMatrixSkel members (baspacho/baspacho/CoalescedBlockMatrix.h:88-110). */
#ifndef ORC_SKEL_H_
#define ORC_SKEL_H_
#include <stdint.h>

typedef struct orc_skel {
  int64_t numSpans, numLumps;
  const int64_t *spanStart, *spanToLump, *lumpStart, *lumpToSpan, *spanOffsetInLump;
  const int64_t *chainColPtr, *chainRowSpan, *chainData, *chainRowsTillEnd;
  const int64_t *boardColPtr, *boardRowLump, *boardChainColOrd;
  const int64_t *boardRowPtr, *boardColLump, *boardColOrd;
} orc_skel;

#endif
