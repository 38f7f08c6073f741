"""``test_score`` with ONE rank made to fail inside its run (tests/test_gpu_parity.py::test_cli_rank_failure_ends_every_rank_quickly):
``python failing_rank_cli.py <rank> <test_score arguments ...>``.  The failure is injected here, by wrapping ``AldBatch.run`` -- the
product carries no test hook for it."""
import os
import sys

if __name__ == '__main__':
    fail_rank = sys.argv[1]
    sys.argv = ['test_score'] + sys.argv[2:]
    from score_based_channels_amd import ald, test_score

    def broken(self, *a, **k):
        raise RuntimeError('rank %s fails before the gather, on request' % fail_rank)
    if os.environ.get('RANK', '0') == fail_rank:
        ald.AldBatch.run = broken
        ald.AldBatch.run_leading = broken
        ald.AldBatch.run_following = broken
    test_score.main()
