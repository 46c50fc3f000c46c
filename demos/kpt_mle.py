"""Counterpart of the reference's tetralith/jobs/kpt_mle.py: `python demos/kpt_mle.py [--num-mcs 100] [--T 3141] [--results DIR]`
(the job table and the loop live in demos/jobs.py)."""
import sys

from jobs import main

if __name__ == '__main__':
    main(['kpt_mle'] + sys.argv[1:])
