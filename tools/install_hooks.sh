#!/bin/bash
# Point this clone's git hooks at tools/git-hooks (the pre-commit size guard).
cd "$(dirname "$0")/.." && ln -sf ../../tools/git-hooks/pre-commit .git/hooks/pre-commit && echo "installed .git/hooks/pre-commit"
